/* owl_host.h -- the host C-ABI of OWL, exported by libowl_mi355x.so.
 *
 * Source-compatible with the reference's owl/include/owl/owl_host.h: same entry-point names,
 * argument lists, enum values and POD layouts, so code written against it (samples/s01-trueknn/
 * hostCode.cpp) compiles and links unchanged.  Each group below cites the reference lines it
 * mirrors.  The ~560 typed variable setters are generated from one table instead of being spelled
 * out.  Entry points whose subsystem is outside the TrueKNN / RT-DBSCAN path (textures, triangle
 * meshes, graphics interop, motion blur) are declared and throw "not supported on this backend".
 *
 * Error convention (reference: owl/helper/cuda.h:22-31, impl.cpp:222-224): no status codes; a
 * failing call throws std::runtime_error through the C boundary.
 */
#ifndef OWL_HOST_H_MI355X
#define OWL_HOST_H_MI355X

#include <cuda.h>          /* shim: CUstream, CUtexObject        (reference :19) */
#include <driver_types.h>  /* shim: cudaGraphicsResource_t        (reference :20) */
#include <optix.h>         /* shim: OptixTraversableHandle, ...   (reference :21) */

#include <stdint.h>
#include <sys/types.h>
#ifdef __cplusplus
#include <cstddef>
#endif

#define OWL_DLL_EXPORT __attribute__((visibility("default")))
#define OWL_DLL_IMPORT __attribute__((visibility("default")))

#ifdef __cplusplus
#define OWL_IF_CPP(a) a
#define OWL_API extern "C" OWL_DLL_EXPORT
#else
#define OWL_IF_CPP(a)
#define OWL_API OWL_DLL_EXPORT
#endif

/* reference :70-73 */
#define OWL_OFFSETOF(type, member) ((char *)(&((struct type *)0)->member) - (char *)(((struct type *)0)))

/* reference :76-96 */
typedef enum {
  OWL_MATRIX_FORMAT_COLUMN_MAJOR = 0,
  OWL_MATRIX_FORMAT_OWL = OWL_MATRIX_FORMAT_COLUMN_MAJOR,
  OWL_MATRIX_FORMAT_ROW_MAJOR
} OWLMatrixFormat;

/* reference :98-105 */
typedef enum {
  OWL_SBT_HITGROUPS = 0x1,
  OWL_SBT_GEOMS = OWL_SBT_HITGROUPS,
  OWL_SBT_RAYGENS = 0x2,
  OWL_SBT_MISSPROGS = 0x4,
  OWL_SBT_ALL = 0x7
} OWLBuildSBTFlags;

/* reference :107-216 -- the numeric values are ABI */
typedef enum {
  OWL_INVALID_TYPE = 0,
  OWL_BUFFER = 10,
  OWL_BUFFER_SIZE,
  OWL_BUFFER_ID,
  OWL_BUFFER_POINTER,
  OWL_BUFPTR = OWL_BUFFER_POINTER,
  OWL_GROUP = 20,
  OWL_DEVICE = 30,
  OWL_TEXTURE = 40,
  OWL_TEXTURE_2D = OWL_TEXTURE,
  _OWL_BEGIN_COPYABLE_TYPES = 1000,
  OWL_FLOAT = 1000, OWL_FLOAT2, OWL_FLOAT3, OWL_FLOAT4,
  OWL_INT = 1010, OWL_INT2, OWL_INT3, OWL_INT4,
  OWL_UINT = 1020, OWL_UINT2, OWL_UINT3, OWL_UINT4,
  OWL_LONG = 1030, OWL_LONG2, OWL_LONG3, OWL_LONG4,
  OWL_ULONG = 1040, OWL_ULONG2, OWL_ULONG3, OWL_ULONG4,
  OWL_DOUBLE = 1050, OWL_DOUBLE2, OWL_DOUBLE3, OWL_DOUBLE4,
  OWL_CHAR = 1060, OWL_CHAR2, OWL_CHAR3, OWL_CHAR4,
  OWL_UCHAR = 1070, OWL_UCHAR2, OWL_UCHAR3, OWL_UCHAR4,
  OWL_SHORT = 1080, OWL_SHORT2, OWL_SHORT3, OWL_SHORT4,
  OWL_USHORT = 1090, OWL_USHORT2, OWL_USHORT3, OWL_USHORT4,
  OWL_BOOL, OWL_BOOL2, OWL_BOOL3, OWL_BOOL4,
  OWL_RAW_POINTER = OWL_ULONG,
  OWL_BYTE = OWL_UCHAR,
  OWL_AFFINE3F = 1300,
  OWL_USER_TYPE_BEGIN = 10000
} OWLDataType;

#define OWL_USER_TYPE(userType) ((OWLDataType)(OWL_USER_TYPE_BEGIN + sizeof(userType)))

/* reference :220-233 */
typedef enum {
  OWL_GEOMETRY_USER,
  OWL_GEOM_USER = OWL_GEOMETRY_USER,
  OWL_GEOMETRY_TRIANGLES,
  OWL_GEOM_TRIANGLES = OWL_GEOMETRY_TRIANGLES,
  OWL_TRIANGLES = OWL_GEOMETRY_TRIANGLES,
  OWL_GEOMETRY_HAIR
} OWLGeomKind;

#define OWL_ALL_RAY_TYPES -1

typedef float OWL_float;
typedef double OWL_double;
typedef int32_t OWL_int;
typedef uint32_t OWL_uint;
typedef int64_t OWL_long;
typedef uint64_t OWL_ulong;

/* reference :245-266 (owl2ui has int32_t members there, kept for layout/ABI fidelity) */
typedef struct _OWL_int2 { int32_t x, y; } owl2i;
typedef struct _OWL_uint2 { int32_t x, y; } owl2ui;
typedef struct _OWL_long2 { int64_t x, y; } owl2l;
typedef struct _OWL_ulong2 { uint64_t x, y; } owl2ul;
typedef struct _OWL_float2 { float x, y; } owl2f;
typedef struct _OWL_double2 { double x, y; } owl2d;
typedef struct _OWL_int3 { int32_t x, y, z; } owl3i;
typedef struct _OWL_uint3 { uint32_t x, y, z; } owl3ui;
typedef struct _OWL_long3 { int64_t x, y, z; } owl3l;
typedef struct _OWL_ulong3 { uint64_t x, y, z; } owl3ul;
typedef struct _OWL_float3 { float x, y, z; } owl3f;
typedef struct _OWL_double3 { double x, y, z; } owl3d;
typedef struct _OWL_int4 { int32_t x, y, z, w; } owl4i;
typedef struct _OWL_uint4 { uint32_t x, y, z, w; } owl4ui;
typedef struct _OWL_long4 { int64_t x, y, z, w; } owl4l;
typedef struct _OWL_ulong4 { uint64_t x, y, z, w; } owl4ul;
typedef struct _OWL_float4 { float x, y, z, w; } owl4f;
typedef struct _OWL_double4 { double x, y, z, w; } owl4d;
typedef struct _OWL_affine3f { owl3f vx, vy, vz, t; } owl4x3f;

/* reference :268-272; a list is terminated by name == NULL when numVars == -1 (impl.cpp:269-289) */
typedef struct _OWLVarDecl {
  const char *name;
  OWLDataType type;
  uint32_t offset;
} OWLVarDecl;

/* reference :276-307 (texture enums: declared for source compatibility only) */
typedef enum { OWL_TEXEL_FORMAT_RGBA8, OWL_TEXEL_FORMAT_RGBA32F, OWL_TEXEL_FORMAT_R8, OWL_TEXEL_FORMAT_R32F } OWLTexelFormat;
typedef enum { OWL_TEXTURE_NEAREST, OWL_TEXTURE_LINEAR } OWLTextureFilterMode;
typedef enum { OWL_TEXTURE_WRAP, OWL_TEXTURE_CLAMP, OWL_TEXTURE_BORDER, OWL_TEXTURE_MIRROR } OWLTextureAddressMode;
typedef enum { OWL_COLOR_SPACE_LINEAR, OWL_COLOR_SPACE_SRGB } OWLTextureColorSpace;

typedef OptixTraversableHandle OWLDeviceTraversable;
typedef struct _OWLDeviceBuffer2D { void *d_pointer; owl2i dims; } OWLDeviceBuffer2D;

/* reference :315-334: opaque handles */
typedef struct _OWLContext *OWLContext;
typedef struct _OWLBuffer *OWLBuffer;
typedef struct _OWLTexture *OWLTexture;
typedef struct _OWLGeom *OWLGeom;
typedef struct _OWLGeomType *OWLGeomType;
typedef struct _OWLVariable *OWLVariable;
typedef struct _OWLModule *OWLModule;
typedef struct _OWLGroup *OWLGroup;
typedef struct _OWLRayGen *OWLRayGen;
typedef struct _OWLMissProg *OWLMissProg;
typedef struct _OWLLaunchParams *OWLLaunchParams, *OWLParams, *OWLGlobals;

/* ---- context, programs, SBT (reference :336-418) ------------------------------------------- */
OWL_API void owlBuildPrograms(OWLContext context);
OWL_API void owlBuildPipeline(OWLContext context);
OWL_API void owlBuildSBT(OWLContext context, OWLBuildSBTFlags flags OWL_IF_CPP(= OWL_SBT_ALL));
OWL_API int32_t owlGetDeviceCount(OWLContext context);
OWL_API OWLContext owlContextCreate(int32_t *requestedDeviceIDs OWL_IF_CPP(= nullptr), int numDevices OWL_IF_CPP(= 0));
OWL_API void owlEnableMotionBlur(OWLContext context);
OWL_API void owlContextSetRayTypeCount(OWLContext context, size_t numRayTypes);
OWL_API void owlSetMaxInstancingDepth(OWLContext context, int32_t maxInstanceDepth);
OWL_API void owlContextDestroy(OWLContext context);
OWL_API CUstream owlContextGetStream(OWLContext context, int deviceID);
OWL_API OptixDeviceContext owlContextGetOptixContext(OWLContext context, int deviceID);
OWL_API OWLModule owlModuleCreate(OWLContext context, const char *ptxCode);

/* ---- object creation (reference :420-612) -------------------------------------------------- */
OWL_API OWLGeom owlGeomCreate(OWLContext context, OWLGeomType type);
OWL_API OWLParams owlParamsCreate(OWLContext context, size_t sizeOfVarStruct, OWLVarDecl *vars, int numVars);
OWL_API OWLRayGen owlRayGenCreate(OWLContext context, OWLModule module, const char *programName,
                                  size_t sizeOfVarStruct, OWLVarDecl *vars, int numVars);
OWL_API OWLMissProg owlMissProgCreate(OWLContext context, OWLModule module, const char *programName,
                                      size_t sizeOfVarStruct, OWLVarDecl *vars, int numVars);
OWL_API void owlMissProgSet(OWLContext context, int rayType, OWLMissProg missProgToUse);
OWL_API OWLGroup owlUserGeomGroupCreate(OWLContext context, size_t numGeometries, OWLGeom *arrayOfChildGeoms);
OWL_API OWLGroup owlTrianglesGeomGroupCreate(OWLContext context, size_t numGeometries, OWLGeom *initValues);
OWL_API OWLGroup owlInstanceGroupCreate(OWLContext context, size_t numInstances,
                                        const OWLGroup *initGroups OWL_IF_CPP(= nullptr),
                                        const uint32_t *initInstanceIDs OWL_IF_CPP(= nullptr),
                                        const float *initTransforms OWL_IF_CPP(= nullptr),
                                        OWLMatrixFormat matrixFormat OWL_IF_CPP(= OWL_MATRIX_FORMAT_OWL));
OWL_API void owlGroupBuildAccel(OWLGroup group);
OWL_API void owlGroupRefitAccel(OWLGroup group);
OWL_API OWLGeomType owlGeomTypeCreate(OWLContext context, OWLGeomKind kind, size_t sizeOfVarStruct,
                                      OWLVarDecl *vars, int numVars);
OWL_API OWLTexture owlTexture2DCreate(OWLContext context, OWLTexelFormat texelFormat, uint32_t size_x,
                                      uint32_t size_y, const void *texels,
                                      OWLTextureFilterMode filterMode OWL_IF_CPP(= OWL_TEXTURE_LINEAR),
                                      OWLTextureAddressMode addressMode OWL_IF_CPP(= OWL_TEXTURE_CLAMP),
                                      OWLTextureColorSpace colorSpace OWL_IF_CPP(= OWL_COLOR_SPACE_LINEAR),
                                      uint32_t linePitchInBytes OWL_IF_CPP(= 0));
OWL_API CUtexObject owlTextureGetObject(OWLTexture texture, int deviceID);
OWL_API void owlTexture2DDestroy(OWLTexture texture);
OWL_API OWLBuffer owlDeviceBufferCreate(OWLContext context, OWLDataType type, size_t count, const void *init);
OWL_API OWLBuffer owlHostPinnedBufferCreate(OWLContext context, OWLDataType type, size_t count);
OWL_API OWLBuffer owlManagedMemoryBufferCreate(OWLContext context, OWLDataType type, size_t count, const void *init);
OWL_API OWLBuffer owlGraphicsBufferCreate(OWLContext context, OWLDataType type, size_t count,
                                          cudaGraphicsResource_t resource);
OWL_API void owlGraphicsBufferMap(OWLBuffer buffer);
OWL_API void owlGraphicsBufferUnmap(OWLBuffer buffer);

/* ---- buffers, launches (reference :631-681) ------------------------------------------------ */
OWL_API const void *owlBufferGetPointer(OWLBuffer buffer, int deviceID);
OWL_API OptixTraversableHandle owlGroupGetTraversable(OWLGroup group, int deviceID);
OWL_API void owlBufferResize(OWLBuffer buffer, size_t newItemCount);
OWL_API void owlBufferDestroy(OWLBuffer buffer);
OWL_API void owlBufferUpload(OWLBuffer buffer, const void *hostPtr, size_t offset OWL_IF_CPP(= 0),
                             size_t numBytes OWL_IF_CPP(= size_t(-1)));
OWL_API void owlRayGenLaunch2D(OWLRayGen rayGen, int dims_x, int dims_y);
OWL_API void owlLaunch2D(OWLRayGen rayGen, int dims_x, int dims_y, OWLParams params);
OWL_API void owlAsyncLaunch2D(OWLRayGen rayGen, int dims_x, int dims_y, OWLParams params);
OWL_API CUstream owlParamsGetCudaStream(OWLParams params, int deviceID);
OWL_API void owlLaunchSync(OWLParams params);

/* ---- geometry (reference :685-778) --------------------------------------------------------- */
OWL_API void owlTrianglesSetVertices(OWLGeom triangles, OWLBuffer vertices, size_t count, size_t stride, size_t offset);
OWL_API void owlTrianglesSetMotionVertices(OWLGeom triangles, size_t numKeys, OWLBuffer *vertexArrays,
                                           size_t count, size_t stride, size_t offset);
OWL_API void owlTrianglesSetIndices(OWLGeom triangles, OWLBuffer indices, size_t count, size_t stride, size_t offset);
OWL_API void owlInstanceGroupSetChild(OWLGroup group, int whichChild, OWLGroup child);
OWL_API void owlInstanceGroupSetTransform(OWLGroup group, int whichChild, const float *floats,
                                          OWLMatrixFormat matrixFormat OWL_IF_CPP(= OWL_MATRIX_FORMAT_OWL));
OWL_API void owlInstanceGroupSetTransforms(OWLGroup group, uint32_t timeStep, const float *floatsForThisStimeStep,
                                           OWLMatrixFormat matrixFormat OWL_IF_CPP(= OWL_MATRIX_FORMAT_OWL));
OWL_API void owlInstanceGroupSetInstanceIDs(OWLGroup group, const uint32_t *instanceIDs);
OWL_API void owlGeomTypeSetClosestHit(OWLGeomType type, int rayType, OWLModule module, const char *progName);
OWL_API void owlGeomTypeSetAnyHit(OWLGeomType type, int rayType, OWLModule module, const char *progName);
OWL_API void owlGeomTypeSetIntersectProg(OWLGeomType type, int rayType, OWLModule module, const char *progName);
OWL_API void owlGeomTypeSetBoundsProg(OWLGeomType type, OWLModule module, const char *progName);
OWL_API void owlGeomSetPrimCount(OWLGeom geom, size_t primCount);

/* ---- releases and variable handles (reference :780-805) ------------------------------------ */
OWL_API void owlGeomRelease(OWLGeom geometry);
OWL_API void owlVariableRelease(OWLVariable variable);
OWL_API void owlModuleRelease(OWLModule module);
OWL_API void owlBufferRelease(OWLBuffer buffer);
OWL_API void owlRayGenRelease(OWLRayGen rayGen);
OWL_API void owlGroupRelease(OWLGroup group);
OWL_API OWLVariable owlGeomGetVariable(OWLGeom geom, const char *varName);
OWL_API OWLVariable owlRayGenGetVariable(OWLRayGen geom, const char *varName);
OWL_API OWLVariable owlMissProgGetVariable(OWLMissProg geom, const char *varName);
OWL_API OWLVariable owlParamsGetVariable(OWLParams object, const char *varName);

/* ---- typed setters (reference :807-1232), generated ----------------------------------------
 * OWL_FOREACH_SCALAR(X): X(suffix, C type) for every scalar family the reference declares;
 * for each:  owlVariableSet{1..4}<s>(var, ...), owlVariableSet{2..4}<s>v(var, const T*), and the
 * same seven forms for RayGen / MissProg / Geom / Params addressed by variable name. */
#define OWL_FOREACH_SCALAR_C(X) \
  X(c, int8_t) X(uc, uint8_t) X(s, int16_t) X(us, uint16_t) X(f, float) X(i, int32_t) X(ui, uint32_t) \
  X(d, double) X(l, int64_t) X(ul, uint64_t)
#ifdef __cplusplus
#define OWL_FOREACH_SCALAR(X) X(b, bool) OWL_FOREACH_SCALAR_C(X)
#else
#define OWL_FOREACH_SCALAR(X) OWL_FOREACH_SCALAR_C(X)
#endif

#define OWL_DECLARE_VARIABLE_SETTERS(sfx, T)                                        \
  OWL_API void owlVariableSet1##sfx(OWLVariable var, T val);                        \
  OWL_API void owlVariableSet2##sfx(OWLVariable var, T x, T y);                     \
  OWL_API void owlVariableSet3##sfx(OWLVariable var, T x, T y, T z);                \
  OWL_API void owlVariableSet4##sfx(OWLVariable var, T x, T y, T z, T w);           \
  OWL_API void owlVariableSet2##sfx##v(OWLVariable var, const T *val);              \
  OWL_API void owlVariableSet3##sfx##v(OWLVariable var, const T *val);              \
  OWL_API void owlVariableSet4##sfx##v(OWLVariable var, const T *val);
OWL_FOREACH_SCALAR(OWL_DECLARE_VARIABLE_SETTERS)

#define OWL_DECLARE_OBJECT_SETTERS_(Obj, Handle, sfx, T)                                         \
  OWL_API void owl##Obj##Set1##sfx(Handle obj, const char *name, T val);                         \
  OWL_API void owl##Obj##Set2##sfx(Handle obj, const char *name, T x, T y);                      \
  OWL_API void owl##Obj##Set3##sfx(Handle obj, const char *name, T x, T y, T z);                 \
  OWL_API void owl##Obj##Set4##sfx(Handle obj, const char *name, T x, T y, T z, T w);            \
  OWL_API void owl##Obj##Set2##sfx##v(Handle obj, const char *name, const T *val);               \
  OWL_API void owl##Obj##Set3##sfx##v(Handle obj, const char *name, const T *val);               \
  OWL_API void owl##Obj##Set4##sfx##v(Handle obj, const char *name, const T *val);
#define OWL_DECLARE_OBJECT_SETTERS(sfx, T)                   \
  OWL_DECLARE_OBJECT_SETTERS_(RayGen, OWLRayGen, sfx, T)     \
  OWL_DECLARE_OBJECT_SETTERS_(MissProg, OWLMissProg, sfx, T) \
  OWL_DECLARE_OBJECT_SETTERS_(Geom, OWLGeom, sfx, T)         \
  OWL_DECLARE_OBJECT_SETTERS_(Params, OWLParams, sfx, T)
OWL_FOREACH_SCALAR(OWL_DECLARE_OBJECT_SETTERS)

OWL_API void owlVariableSetGroup(OWLVariable variable, OWLGroup value);
OWL_API void owlVariableSetTexture(OWLVariable variable, OWLTexture value);
OWL_API void owlVariableSetBuffer(OWLVariable variable, OWLBuffer value);
OWL_API void owlVariableSetRaw(OWLVariable variable, const void *valuePtr);
OWL_API void owlVariableSetPointer(OWLVariable variable, const void *valuePtr);

#define OWL_DECLARE_OBJECT_REF_SETTERS(Obj, Handle)                                   \
  OWL_API void owl##Obj##SetTexture(Handle obj, const char *name, OWLTexture val);    \
  OWL_API void owl##Obj##SetPointer(Handle obj, const char *name, const void *val);   \
  OWL_API void owl##Obj##SetBuffer(Handle obj, const char *name, OWLBuffer val);      \
  OWL_API void owl##Obj##SetGroup(Handle obj, const char *name, OWLGroup val);        \
  OWL_API void owl##Obj##SetRaw(Handle obj, const char *name, const void *val);
OWL_DECLARE_OBJECT_REF_SETTERS(RayGen, OWLRayGen)
OWL_DECLARE_OBJECT_REF_SETTERS(Geom, OWLGeom)
OWL_DECLARE_OBJECT_REF_SETTERS(Params, OWLParams)
OWL_DECLARE_OBJECT_REF_SETTERS(MissProg, OWLMissProg)

#ifdef __cplusplus
/* C++ conveniences taking the owl2i/owl3f/... PODs (reference :1240-1345) */
#define OWL_CPP_POD_SETTERS(Obj, Handle, sfx, P2, P3, P4)                                         \
  inline void owl##Obj##Set2##sfx(Handle obj, const char *name, const P2 &v) { owl##Obj##Set2##sfx(obj, name, v.x, v.y); } \
  inline void owl##Obj##Set3##sfx(Handle obj, const char *name, const P3 &v) { owl##Obj##Set3##sfx(obj, name, v.x, v.y, v.z); } \
  inline void owl##Obj##Set4##sfx(Handle obj, const char *name, const P4 &v) { owl##Obj##Set4##sfx(obj, name, v.x, v.y, v.z, v.w); }
#define OWL_CPP_POD_SETTERS_ALL(Obj, Handle)                  \
  OWL_CPP_POD_SETTERS(Obj, Handle, i, owl2i, owl3i, owl4i)    \
  OWL_CPP_POD_SETTERS(Obj, Handle, ui, owl2ui, owl3ui, owl4ui) \
  OWL_CPP_POD_SETTERS(Obj, Handle, f, owl2f, owl3f, owl4f)
OWL_CPP_POD_SETTERS_ALL(Params, OWLParams)
OWL_CPP_POD_SETTERS_ALL(Geom, OWLGeom)
OWL_CPP_POD_SETTERS_ALL(MissProg, OWLMissProg)
OWL_CPP_POD_SETTERS_ALL(RayGen, OWLRayGen)

inline void owlInstanceGroupSetTransform(OWLGroup group, int childID, const owl4x3f &xfm) {
  owlInstanceGroupSetTransform(group, childID, (const float *)&xfm, OWL_MATRIX_FORMAT_OWL);
}
inline void owlInstanceGroupSetTransform(OWLGroup group, int childID, const owl4x3f *xfm) {
  owlInstanceGroupSetTransform(group, childID, (const float *)xfm, OWL_MATRIX_FORMAT_OWL);
}
#endif

#endif /* OWL_HOST_H_MI355X */
