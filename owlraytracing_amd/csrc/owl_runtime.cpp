// owl_runtime.cpp -- the host half of the OWL C-ABI (include/owl/owl_host.h) on MI355X.
//
// Replaces, for the TrueKNN / RT-DBSCAN path, what the reference spreads over owl/impl.cpp,
// APIContext/APIHandle, Context/DeviceContext, Object/RegisteredObject, SBTObject/Variable,
// Module, RayGen/MissProg/LaunchParams, Buffer, Geometry/UserGeom, Group/UserGeomGroup/
// InstanceGroup.  It is a re-design, not a translation: there is
// no OptiX pipeline to assemble, so "programs" are symbols of a gfx950 code object loaded with
// hipModuleLoadData, the shader binding table is a plain device array of GeomRecord
// (include/owl/device_runtime.h), and the acceleration structure is the HIP LBVH of lbvh.hip.
//
// Behaviour kept from the reference (file:line there):
//   * handles are heap objects owning a shared reference; wrong-kind handles throw
//     (owl/APIHandle.h:57-72); owlContextDestroy frees everything (owl/APIContext.cpp:47-65)
//   * variables are declared by (name, type, offset) lists, terminated by name==NULL when
//     numVars==-1 (impl.cpp:269-289); unknown names and type mismatches throw
//     (impl.cpp:222-224, Variable.cpp:27-36); values are materialised when records are written
//     (SBTObject.cpp:113-120): BUFPTR -> device pointer, GROUP -> traversable, DEVICE -> index
//   * owlBuildPrograms must precede owlGroupBuildAccel on a user group (UserGeom.cu:213-216);
//     refit before build throws (UserGeomGroup.cpp:75-76)
//   * launch params are re-marshalled at every launch into the module's `optixLaunchParams`
//     symbol (RayGen.cpp:157-158, DeviceContext.cpp:292); owlLaunch2D = async launch + sync
//     (impl.cpp:168-176)
//   * errors: std::runtime_error thrown through the C boundary (helper/cuda.h:22-31)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "lbvh.h"
#include "owl/owl_host.h"

// device-visible record layouts (kept in sync with include/owl/device_runtime.h, which is
// device-only code; static_asserts below pin the sizes)
namespace rec {
constexpr int kMaxRayTypes = 4;
constexpr int kRaygenBlock = 256;
struct GeomRecord {
  uint64_t intersect[kMaxRayTypes];
  uint64_t closest_hit[kMaxRayTypes];
  uint64_t any_hit[kMaxRayTypes];
  const void *data;
  uint32_t prim_begin;
  uint32_t prim_count;
};
struct AccelHeader {
  uint32_t kind;
  uint32_t count;
};
struct UserGroupAccel {
  AccelHeader h;
  LbvhView bvh;
  const GeomRecord *geoms;
};
struct Instance {
  float o2w[12];
  float w2o[12];
  uint64_t child;
  uint32_t instance_id;
  uint32_t identity;
};
struct InstanceGroupAccel {
  AccelHeader h;
  const Instance *instances;
};
struct MissRecord {
  uint64_t prog;
  const void *data;
};
struct LaunchDesc {
  uint32_t dims[3];
  uint32_t num_miss;
  const void *raygen_data;
  const MissRecord *miss;
  const int32_t *order;  // (device_runtime.h)
};
struct DeviceBufferVar {  // what an OWL_BUFFER variable expands to (owl_device_buffer.h)
  int32_t type;
  int32_t pad;
  uint64_t count;
  const void *data;
};
static_assert(sizeof(GeomRecord) == 3 * 8 * kMaxRayTypes + 16, "GeomRecord layout");
static_assert(sizeof(Instance) == 112, "Instance layout");
static_assert(sizeof(LaunchDesc) == 40, "LaunchDesc layout");
}  // namespace rec

namespace {

[[noreturn]] void fail(const std::string &msg) { throw std::runtime_error("owl(mi355x): " + msg); }

#define OWL_HIP(call)                                                                             \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) fail(std::string(#call) + " failed: " + hipGetErrorString(e_));         \
  } while (0)

size_t size_of_type(OWLDataType t) {
  if (t >= OWL_USER_TYPE_BEGIN) return (size_t)t - OWL_USER_TYPE_BEGIN;
  switch (t) {
    case OWL_BUFFER: return sizeof(rec::DeviceBufferVar);
    case OWL_BUFFER_SIZE: return 8;
    case OWL_BUFFER_ID: return 4;
    case OWL_BUFFER_POINTER: return 8;
    case OWL_GROUP: return 8;
    case OWL_DEVICE: return 4;
    case OWL_TEXTURE: return 8;
    case OWL_AFFINE3F: return 48;
    default: break;
  }
  if (t >= OWL_FLOAT && t <= OWL_BOOL4) {
    const int family = ((int)t - 1000) / 10, n = ((int)t - 1000) % 10 + 1;
    static const size_t scalar[] = {4, 4, 4, 8, 8, 8, 1, 1, 2, 2};  // f i ui l ul d c uc s us
    if (family <= 9 && n <= 4) return scalar[family] * n;
    if (t >= OWL_BOOL && t <= OWL_BOOL4) return (size_t)((int)t - (int)OWL_BOOL + 1);
  }
  fail("sizeOf: unsupported OWLDataType " + std::to_string((int)t));
}

struct Context;

enum class Kind { Context, Module, GeomType, Geom, Buffer, Group, RayGen, MissProg, Params, Variable, Texture };
const char *kind_name(Kind k) {
  static const char *n[] = {"Context", "Module", "GeomType", "Geom", "Buffer", "Group", "RayGen", "MissProg",
                            "Params", "Variable", "Texture"};
  return n[(int)k];
}

struct Object : std::enable_shared_from_this<Object> {
  Object(Context *c, Kind k) : ctx(c), kind(k) {}
  virtual ~Object() {}
  Context *ctx;
  Kind kind;
};

// every owl*Create / owl*GetVariable returns one of these (reference: APIHandle)
struct Handle {
  std::shared_ptr<Object> obj;
  Context *ctx;
};

struct VarDecl {
  std::string name;
  OWLDataType type;
  uint32_t offset;
};

struct Buffer;
struct Group;

// an object with a typed variable struct (geom, raygen, miss prog, launch params)
struct SBTObject : Object {
  SBTObject(Context *c, Kind k, size_t bytes, const std::vector<VarDecl> *d) : Object(c, k), decls(d), host(bytes, 0) {}
  const std::vector<VarDecl> *decls;
  std::vector<uint8_t> host;                              // plain-data members live here
  std::map<int, std::shared_ptr<Buffer>> buffer_refs;     // decl index -> buffer
  std::map<int, std::shared_ptr<Group>> group_refs;
  int find(const char *name) const {
    if (!name) fail("variable name is NULL");
    for (size_t i = 0; i < decls->size(); i++)
      if ((*decls)[i].name == name) return (int)i;
    fail(std::string("could not find variable '") + name + "'");
  }
  void materialise(std::vector<uint8_t> &out) const;  // resolves buffers/groups to device values
};

struct Module : Object {
  Module(Context *c, const char *code_) : Object(c, Kind::Module), code(code_) {}
  ~Module() override {
    if (params_free) (void)hipEventDestroy(params_free);
    if (mod) (void)hipModuleUnload(mod);
  }
  const char *code;
  hipModule_t mod = nullptr;
  hipDeviceptr_t params_ptr = nullptr;  // `optixLaunchParams`, if the module defines it
  size_t params_bytes = 0;
  // `optixLaunchParams` is ONE global of the user's code object, while every OWLParams launches on a
  // stream of its own (reference: a device buffer per LaunchParams, LaunchParams.cpp:38-49, so async
  // launches with different params may overlap there).  Here the next upload into the global waits
  // for the last kernel that reads it, whatever stream that ran on.
  hipEvent_t params_free = nullptr;
  hipStream_t params_stream = nullptr;
  bool params_in_use = false;
  void load() {
    if (mod) return;
    if (!code) fail("owlModuleCreate was given a NULL code pointer");
    OWL_HIP(hipModuleLoadData(&mod, code));
    {
      // the device header's layout word against this library's (include/owl/device_runtime.h, OWL_MI355X_DEVICE_ABI)
      hipDeviceptr_t word = nullptr;
      size_t bytes = 0;
      unsigned theirs = 0;
      const unsigned mine = (2u << 16) | (unsigned)sizeof(rec::LaunchDesc);
      if (hipModuleGetGlobal(&word, &bytes, mod, "owl_mi355x_device_abi") != hipSuccess || bytes != sizeof(unsigned)) {
        (void)hipGetLastError();
        fail("the module carries no owl_mi355x_device_abi word: it was built with another (older) owl/device_runtime.h than this library's");
      }
      OWL_HIP(hipMemcpy(&theirs, word, sizeof theirs, hipMemcpyDeviceToHost));
      if (theirs != mine)
        fail("device code was built with another revision of owl/device_runtime.h (layout word " + std::to_string(theirs) + ", this library's is " +
             std::to_string(mine) + "): rebuild the device programs against this library's headers");
    }
    if (hipModuleGetGlobal(&params_ptr, &params_bytes, mod, "optixLaunchParams") != hipSuccess) {
      (void)hipGetLastError();
      params_ptr = nullptr;
      params_bytes = 0;
    }
  }
  hipFunction_t kernel(const std::string &name) {
    hipFunction_t f = nullptr;
    if (hipModuleGetFunction(&f, mod, name.c_str()) != hipSuccess) {
      (void)hipGetLastError();
      fail("kernel '" + name + "' not found in module (compile device code with owl/owl_device.h for gfx950)");
    }
    return f;
  }
  uint64_t program_pointer(const std::string &symbol) {  // value of __owl_fp__<symbol>
    hipDeviceptr_t p = nullptr;
    size_t bytes = 0;
    const std::string var = "__owl_fp__" + symbol;
    if (hipModuleGetGlobal(&p, &bytes, mod, var.c_str()) != hipSuccess || bytes != 8) {
      (void)hipGetLastError();
      fail("program '" + symbol + "' not found in module");
    }
    uint64_t v = 0;
    OWL_HIP(hipMemcpy(&v, p, 8, hipMemcpyDeviceToHost));
    return v;
  }
};

struct ProgRef {
  std::shared_ptr<Module> module;
  std::string name;
  uint64_t fn = 0;
};

struct GeomType : Object {
  GeomType(Context *c, OWLGeomKind k, size_t bytes, std::vector<VarDecl> d)
      : Object(c, Kind::GeomType), geom_kind(k), var_bytes(bytes), decls(std::move(d)) {}
  OWLGeomKind geom_kind;
  size_t var_bytes;
  std::vector<VarDecl> decls;
  ProgRef intersect[rec::kMaxRayTypes], closest_hit[rec::kMaxRayTypes], any_hit[rec::kMaxRayTypes];
  ProgRef bounds;
  hipFunction_t bounds_kernel = nullptr;
};

struct DeviceBlob {  // small owned device allocation
  void *ptr = nullptr;
  size_t bytes = 0;
  ~DeviceBlob() { release(); }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
  }
  void reserve(size_t n) {
    if (n <= bytes && ptr) return;
    release();
    OWL_HIP(hipMalloc(&ptr, n ? n : 16));
    bytes = n ? n : 16;
  }
  void upload(const void *src, size_t n, hipStream_t s) {
    reserve(n);
    if (n) OWL_HIP(hipMemcpyAsync(ptr, src, n, hipMemcpyHostToDevice, s));
  }
};

struct Geom : SBTObject {
  Geom(Context *c, std::shared_ptr<GeomType> t)
      : SBTObject(c, Kind::Geom, t->var_bytes, &t->decls), type(std::move(t)) {}
  std::shared_ptr<GeomType> type;
  size_t prim_count = 0;
  DeviceBlob data;  // the variable struct on the device = SBT data of this geometry
};

enum class BufferKind { Device, Managed, HostPinned };
struct Buffer : Object {
  Buffer(Context *c, BufferKind bk, OWLDataType t, size_t n) : Object(c, Kind::Buffer), bkind(bk), type(t), count(n) {
    static int created = 0;
    serial = created++;
  }
  int serial;  // in the order of creation (the dump hook of owlContextDestroy names its files by it)
  ~Buffer() override { release(); }
  BufferKind bkind;
  OWLDataType type;
  size_t count;
  void *ptr = nullptr;
  // Managed buffers in "mirror" form (the default, see managed_policy): `ptr` is DEVICE memory -- what BUFPTR
  // variables hand to device code -- and `mirror` a pinned host copy -- what owlBufferGetPointer hands to host code.
  // Both sides see one coherent buffer at the API's synchronisation points (owlLaunch2D's return, owlLaunchSync,
  // owlBufferGetPointer), like managed memory after a device synchronisation: once the host has been given the
  // pointer it may keep it, so from then on every launch is preceded by a mirror -> device copy and every
  // synchronisation point followed by a device -> mirror copy (refresh_mirrors) -- a host program that fetches the
  // pointer once and re-reads it after later launches sees their results (tests/test_owl_api.py).
  void *mirror = nullptr;
  bool mirrored = false;      // this buffer uses the mirror form
  bool device_newer = false;  // a launch ran since the mirror was last refreshed
  bool handed_out = false;    // the host holds the mirror's pointer (it may read and write it between synchronisation points)
  size_t elem() const { return size_of_type(type); }
  size_t bytes() const { return elem() * count; }
  void release() {
    if (mirror) (void)hipHostFree(mirror);
    mirror = nullptr;
    if (!ptr) return;
    if (bkind == BufferKind::HostPinned)
      (void)hipHostFree(ptr);
    else
      (void)hipFree(ptr);
    ptr = nullptr;
  }
  const void *host_pointer();  // owlBufferGetPointer
  void write_back(hipStream_t s);  // before a launch
  void allocate(const void *init);
  void resize(size_t n) {
    release();
    count = n;
    allocate(nullptr);
  }
};

struct Group : Object {
  Group(Context *c, bool inst) : Object(c, Kind::Group), is_instance(inst) {}
  bool is_instance;
  bool built = false;
  DeviceBlob accel;  // rec::UserGroupAccel or rec::InstanceGroupAccel
  virtual void build(bool refit) = 0;
  uint64_t traversable() const { return (uint64_t)accel.ptr; }
};

struct UserGeomGroup : Group {
  UserGeomGroup(Context *c, std::vector<std::shared_ptr<Geom>> g) : Group(c, false), geoms(std::move(g)) {
    accel.reserve(sizeof(rec::UserGroupAccel));  // address is stable from creation on
  }
  std::vector<std::shared_ptr<Geom>> geoms;
  owlmi::Lbvh bvh;
  DeviceBlob boxes, records;
  void build(bool refit) override;
  void write_records(hipStream_t s);
};

struct InstanceGroup : Group {
  InstanceGroup(Context *c, size_t n) : Group(c, true), children(n), ids(n), xfms(n) {
    accel.reserve(sizeof(rec::InstanceGroupAccel));
    for (size_t i = 0; i < n; i++) {
      ids[i] = (uint32_t)i;
      float *m = xfms[i].m;
      std::memset(m, 0, sizeof(float) * 12);
      m[0] = m[5] = m[10] = 1.f;
    }
  }
  struct Xfm {
    float m[12];  // row-major 3x4 object-to-world
  };
  std::vector<std::shared_ptr<Group>> children;
  std::vector<uint32_t> ids;
  std::vector<Xfm> xfms;
  DeviceBlob instances;
  void set_transform(size_t i, const float *f, OWLMatrixFormat fmt) {
    if (i >= xfms.size()) fail("instance index out of range");
    float *m = xfms[i].m;
    if (fmt == OWL_MATRIX_FORMAT_ROW_MAJOR) {
      std::memcpy(m, f, sizeof(float) * 12);
    } else {  // OWL: vx, vy, vz, t as columns
      for (int r = 0; r < 3; r++) {
        m[4 * r + 0] = f[0 + r];
        m[4 * r + 1] = f[3 + r];
        m[4 * r + 2] = f[6 + r];
        m[4 * r + 3] = f[9 + r];
      }
    }
  }
  void build(bool refit) override;
};

struct RayGen : SBTObject {
  RayGen(Context *c, std::shared_ptr<Module> m, std::string n, size_t bytes, std::vector<VarDecl> d)
      : SBTObject(c, Kind::RayGen, bytes, &own_decls), module(std::move(m)), name(std::move(n)), own_decls(std::move(d)) {}
  std::shared_ptr<Module> module;
  std::string name;
  std::vector<VarDecl> own_decls;
  hipFunction_t kernel = nullptr;
  DeviceBlob data;
};

struct MissProg : SBTObject {
  MissProg(Context *c, std::shared_ptr<Module> m, std::string n, size_t bytes, std::vector<VarDecl> d)
      : SBTObject(c, Kind::MissProg, bytes, &own_decls), prog{std::move(m), std::move(n), 0}, own_decls(std::move(d)) {}
  ProgRef prog;
  std::vector<VarDecl> own_decls;
  DeviceBlob data;
};

struct Params : SBTObject {
  Params(Context *c, size_t bytes, std::vector<VarDecl> d) : SBTObject(c, Kind::Params, bytes, &own_decls), own_decls(std::move(d)) {
    OWL_HIP(hipStreamCreate(&stream));
  }
  ~Params() override {
    if (stream) (void)hipStreamDestroy(stream);
  }
  std::vector<VarDecl> own_decls;
  hipStream_t stream = nullptr;
  std::vector<uint8_t> staging;  // outlives the async upload of the current launch
};

struct Variable : Object {
  Variable(Context *c, std::shared_ptr<SBTObject> o, int i) : Object(c, Kind::Variable), owner(std::move(o)), index(i) {}
  std::shared_ptr<SBTObject> owner;
  int index;
};

struct Context : Object {
  Context() : Object(nullptr, Kind::Context) { ctx = this; }
  int device = 0;
  hipStream_t stream = nullptr;
  size_t num_ray_types = 1;
  std::mutex mtx;                   // guards `handles` (reference: APIContext monitor)
  std::set<Handle *> handles;
  std::vector<std::weak_ptr<Module>> modules;
  std::vector<std::weak_ptr<GeomType>> geom_types;
  std::vector<std::weak_ptr<Geom>> geoms;
  std::vector<std::weak_ptr<RayGen>> raygens;
  std::vector<std::weak_ptr<MissProg>> miss_progs;
  std::vector<std::weak_ptr<Buffer>> buffers;
  std::vector<std::weak_ptr<UserGeomGroup>> user_groups;
  std::vector<std::shared_ptr<MissProg>> miss_by_ray_type;
  DeviceBlob miss_records;
  bool programs_built = false;

  Handle *make_handle(std::shared_ptr<Object> o) {
    Handle *h = new Handle{std::move(o), this};
    std::lock_guard<std::mutex> g(mtx);
    handles.insert(h);
    return h;
  }
  void drop_handle(Handle *h) {
    {
      std::lock_guard<std::mutex> g(mtx);
      handles.erase(h);
    }
    delete h;
  }
};

template <typename T>
std::shared_ptr<T> get(const void *handle, Kind want) {
  if (!handle) fail(std::string("NULL handle where a ") + kind_name(want) + " was expected");
  const Handle *h = (const Handle *)handle;
  if (!h->obj) fail("handle was already released");
  if (h->obj->kind != want)
    fail(std::string("handle of kind ") + kind_name(h->obj->kind) + " used where a " + kind_name(want) + " was expected");
  return std::static_pointer_cast<T>(h->obj);
}
std::shared_ptr<SBTObject> get_sbt(const void *handle, Kind want) { return std::static_pointer_cast<SBTObject>(get<Object>(handle, want)); }

std::vector<VarDecl> copy_decls(const OWLVarDecl *vars, int num) {
  std::vector<VarDecl> out;
  if (!vars) return out;
  if (num < 0) {
    for (int i = 0; vars[i].name; i++) out.push_back({vars[i].name, vars[i].type, vars[i].offset});
  } else {
    for (int i = 0; i < num; i++) {
      if (!vars[i].name) fail("variable declaration without a name");
      out.push_back({vars[i].name, vars[i].type, vars[i].offset});
    }
  }
  return out;
}

void check_decls(const std::vector<VarDecl> &decls, size_t bytes) {
  for (const VarDecl &d : decls)
    if ((size_t)d.offset + size_of_type(d.type) > bytes)
      fail("variable '" + d.name + "' does not fit the declared struct size");
}

int managed_policy();

void Buffer::allocate(const void *init) {
  const size_t n = bytes();
  if (bkind == BufferKind::Device) {
    OWL_HIP(hipMalloc(&ptr, n ? n : 16));
    if (init && n) OWL_HIP(hipMemcpy(ptr, init, n, hipMemcpyHostToDevice));
  } else if (bkind == BufferKind::Managed && (managed_policy() == 4 || managed_policy() == 5)) {
    // mirror form: device memory + pinned host copy (one address per side; see struct Buffer)
    mirrored = true;
    OWL_HIP(hipMalloc(&ptr, n ? n : 16));
    OWL_HIP(hipHostMalloc(&mirror, n ? n : 16, hipHostMallocDefault));
    if (init && n) {
      std::memcpy(mirror, init, n);
      OWL_HIP(hipMemcpy(ptr, mirror, n, hipMemcpyHostToDevice));
    }
    device_newer = false;
    handed_out = false;
  } else if (bkind == BufferKind::Managed) {
    // one address valid on host and device (owlBufferGetPointer is read on the host by
    // samples/s01-trueknn/hostCode.cpp:294-313)
    mirrored = false;
    OWL_HIP(hipMallocManaged(&ptr, n ? n : 16, hipMemAttachGlobal));
    if (init && n) std::memcpy(ptr, init, n);
    if (managed_policy() != 2 && managed_policy() != 3) (void)hipMemAdvise(ptr, n ? n : 16, hipMemAdviseSetPreferredLocation, ctx->device);
    if (hipMemPrefetchAsync(ptr, n ? n : 16, ctx->device, ctx->stream) != hipSuccess) (void)hipGetLastError();
    OWL_HIP(hipStreamSynchronize(ctx->stream));
  } else {
    OWL_HIP(hipHostMalloc(&ptr, n ? n : 16, hipHostMallocDefault));
    if (init && n) std::memcpy(ptr, init, n);
  }
}

void SBTObject::materialise(std::vector<uint8_t> &out) const {
  out = host;
  for (size_t i = 0; i < decls->size(); i++) {
    const VarDecl &d = (*decls)[i];
    uint8_t *dst = out.data() + d.offset;
    switch (d.type) {
      case OWL_BUFFER_POINTER: {
        auto it = buffer_refs.find((int)i);
        const void *p = it == buffer_refs.end() ? nullptr : it->second->ptr;
        std::memcpy(dst, &p, 8);
        break;
      }
      case OWL_BUFFER: {
        rec::DeviceBufferVar v = {0, 0, 0, nullptr};
        auto it = buffer_refs.find((int)i);
        if (it != buffer_refs.end()) {
          v.type = (int32_t)it->second->type;
          v.count = it->second->count;
          v.data = it->second->ptr;
        }
        std::memcpy(dst, &v, sizeof v);
        break;
      }
      case OWL_BUFFER_SIZE: {
        auto it = buffer_refs.find((int)i);
        uint64_t n = it == buffer_refs.end() ? 0 : it->second->count;
        std::memcpy(dst, &n, 8);
        break;
      }
      case OWL_GROUP: {
        auto it = group_refs.find((int)i);
        uint64_t t = it == group_refs.end() ? 0 : it->second->traversable();
        std::memcpy(dst, &t, 8);
        break;
      }
      case OWL_DEVICE: {
        int32_t zero = 0;  // one device per context
        std::memcpy(dst, &zero, 4);
        break;
      }
      default: break;  // plain data already in `host`
    }
  }
}

void launch_kernel(hipFunction_t f, uint64_t n_threads, unsigned block, void **args, hipStream_t s) {
  const uint64_t blocks = (n_threads + block - 1) / block;
  if (blocks == 0) return;
  if (blocks > 0x7fffffffull) fail("launch too large");
  OWL_HIP(hipModuleLaunchKernel(f, (unsigned)blocks, 1, 1, block, 1, 1, 0, s, args, nullptr));
}

void UserGeomGroup::write_records(hipStream_t s) {
  std::vector<rec::GeomRecord> recs(geoms.size());
  uint32_t begin = 0;
  for (size_t g = 0; g < geoms.size(); g++) {
    Geom &geom = *geoms[g];
    std::vector<uint8_t> blob;
    geom.materialise(blob);
    geom.data.upload(blob.data(), blob.size(), s);
    rec::GeomRecord &r = recs[g];
    std::memset(&r, 0, sizeof r);
    for (int t = 0; t < rec::kMaxRayTypes; t++) {
      r.intersect[t] = geom.type->intersect[t].fn;
      r.closest_hit[t] = geom.type->closest_hit[t].fn;
      r.any_hit[t] = geom.type->any_hit[t].fn;
    }
    r.data = geom.data.ptr;
    r.prim_begin = begin;
    r.prim_count = (uint32_t)geom.prim_count;
    begin += (uint32_t)geom.prim_count;
  }
  records.upload(recs.data(), recs.size() * sizeof(rec::GeomRecord), s);
}

void UserGeomGroup::build(bool refit) {
  hipStream_t s = ctx->stream;
  if (refit && !built) fail("owlGroupRefitAccel: group was never built (refit before build)");
  uint64_t total = 0;
  for (auto &g : geoms) total += g->prim_count;
  if (total == 0) fail("user geometry group has no primitives");
  if (total >= 0x7fffffffull) fail("too many primitives in one group");
  boxes.reserve(total * sizeof(LbvhBox));
  write_records(s);  // also refreshes each geometry's variable struct (e.g. a changed radius)
  uint64_t begin = 0;
  for (auto &g : geoms) {
    GeomType &t = *g->type;
    if (t.bounds.name.empty()) fail("geometry type has no bounds program (owlGeomTypeSetBoundsProg)");
    if (!t.bounds_kernel) fail("bounds kernel set, but not yet compiled - did you forget to call owlBuildPrograms() before owlGroupBuildAccel()?");
    const void *geom_data = g->data.ptr;
    void *out = (char *)boxes.ptr + begin * sizeof(LbvhBox);
    uint32_t n = (uint32_t)g->prim_count;
    void *args[] = {(void *)&geom_data, (void *)&out, (void *)&n};
    launch_kernel(t.bounds_kernel, n, 256, args, s);
    begin += g->prim_count;
  }
  try {
    if (refit)
      bvh.refit_boxes((const LbvhBox *)boxes.ptr, s);
    else
      bvh.build_from_boxes((const LbvhBox *)boxes.ptr, (int64_t)total, s);
  } catch (const owlmi::HipError &e) {
    fail(e.what);
  }
  rec::UserGroupAccel a;
  a.h.kind = 1;
  a.h.count = (uint32_t)geoms.size();
  a.bvh = bvh.view();
  a.geoms = (const rec::GeomRecord *)records.ptr;
  accel.upload(&a, sizeof a, s);
  OWL_HIP(hipStreamSynchronize(s));  // builds are synchronous in the reference (CUDA_SYNC_CHECK)
  built = true;
}

void invert_3x4(const float *m, float *inv) {
  const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
  const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
  const double id = det != 0.0 ? 1.0 / det : 0.0;
  const double r[9] = {(e * i - f * h) * id, (c * h - b * i) * id, (b * f - c * e) * id,
                       (f * g - d * i) * id, (a * i - c * g) * id, (c * d - a * f) * id,
                       (d * h - e * g) * id, (b * g - a * h) * id, (a * e - b * d) * id};
  for (int row = 0; row < 3; row++) {
    for (int col = 0; col < 3; col++) inv[4 * row + col] = (float)r[3 * row + col];
    inv[4 * row + 3] = (float)-(r[3 * row] * m[3] + r[3 * row + 1] * m[7] + r[3 * row + 2] * m[11]);
  }
}

void InstanceGroup::build(bool refit) {
  hipStream_t s = ctx->stream;
  if (refit && !built) fail("owlGroupRefitAccel: group was never built (refit before build)");
  std::vector<rec::Instance> inst(children.size());
  static const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  for (size_t n = 0; n < children.size(); n++) {
    rec::Instance &r = inst[n];
    std::memcpy(r.o2w, xfms[n].m, sizeof r.o2w);
    invert_3x4(r.o2w, r.w2o);
    r.identity = std::memcmp(r.o2w, ident, sizeof ident) == 0;
    r.instance_id = ids[n];
    r.child = 0;
    if (children[n]) {
      if (!children[n]->built) fail("instance group built before its child group (call owlGroupBuildAccel on the child first)");
      r.child = children[n]->traversable();
    }
  }
  instances.upload(inst.data(), inst.size() * sizeof(rec::Instance), s);
  rec::InstanceGroupAccel a;
  a.h.kind = 2;
  a.h.count = (uint32_t)inst.size();
  a.instances = (const rec::Instance *)instances.ptr;
  accel.upload(&a, sizeof a, s);
  OWL_HIP(hipStreamSynchronize(s));
  built = true;
}

template <typename T>
void prune(std::vector<std::weak_ptr<T>> &v) {
  size_t w = 0;
  for (auto &p : v)
    if (!p.expired()) v[w++] = p;
  v.resize(w);
}

void resolve(ProgRef &p, const char *prefix) {
  if (p.name.empty() || !p.module) return;
  p.module->load();
  p.fn = p.module->program_pointer(std::string(prefix) + p.name);
}

void build_programs(Context &c) {
  OWL_HIP(hipSetDevice(c.device));
  prune(c.modules);
  prune(c.geom_types);
  prune(c.raygens);
  prune(c.miss_progs);
  for (auto &wm : c.modules)
    if (auto m = wm.lock()) m->load();
  for (auto &wt : c.geom_types) {
    auto t = wt.lock();
    if (!t) continue;
    for (int r = 0; r < rec::kMaxRayTypes; r++) {
      resolve(t->intersect[r], "__intersection__");
      resolve(t->closest_hit[r], "__closesthit__");
      resolve(t->any_hit[r], "__anyhit__");
    }
    if (!t->bounds.name.empty() && t->bounds.module) {
      t->bounds.module->load();
      t->bounds_kernel = t->bounds.module->kernel("__boundsFuncKernel__" + t->bounds.name);
    }
  }
  for (auto &wr : c.raygens)
    if (auto r = wr.lock()) {
      r->module->load();
      r->kernel = r->module->kernel("__raygen__" + r->name);
    }
  for (auto &wm : c.miss_progs)
    if (auto m = wm.lock()) resolve(m->prog, "__miss__");
  c.programs_built = true;
}

void build_sbt(Context &c, int flags) {
  hipStream_t s = c.stream;
  if (flags & OWL_SBT_HITGROUPS) {
    prune(c.user_groups);
    for (auto &wg : c.user_groups)
      if (auto g = wg.lock()) g->write_records(s);
  }
  if (flags & OWL_SBT_RAYGENS) {
    prune(c.raygens);
    for (auto &wr : c.raygens)
      if (auto r = wr.lock()) {
        std::vector<uint8_t> blob;
        r->materialise(blob);
        r->data.upload(blob.data(), blob.size(), s);
      }
  }
  if (flags & OWL_SBT_MISSPROGS) {
    std::vector<rec::MissRecord> recs(c.miss_by_ray_type.size());
    for (size_t i = 0; i < recs.size(); i++) {
      recs[i] = {0, nullptr};
      if (auto &m = c.miss_by_ray_type[i]) {
        std::vector<uint8_t> blob;
        m->materialise(blob);
        m->data.upload(blob.data(), blob.size(), s);
        recs[i].prog = m->prog.fn;
        recs[i].data = m->data.ptr;
      }
    }
    c.miss_records.upload(recs.data(), recs.size() * sizeof(rec::MissRecord), s);
  }
  OWL_HIP(hipStreamSynchronize(s));
}

// How "managed" buffers (owlManagedMemoryBufferCreate) are kept coherent between launches and the host code that reads
// them -- samples/s01-trueknn/hostCode.cpp:294-330 scans its n*k*24-byte frameBuffer after every launch.
//   4 (default) mirror form: device memory for device code, a pinned host copy for owlBufferGetPointer, one bulk copy
//     each way at the API's synchronisation points (struct Buffer).  The sample at 1 M points, k = 10: three rounds in
//     about a quarter of the time of policy 0 (profiles/).
//   5 (opt-in) as 4, for applications whose HOST code only reads its managed buffers (the TrueKNN sample's loop does): no
//     mirror -> device copy before a launch -- what the host writes through the pointer never reaches the device
//     (owlBufferUpload still does).  The sample's three rounds at 1 M points: 57 -> 46 ms.
//   0: hipMallocManaged, prefetched to the GPU before a launch, pulled back by the host's page faults (round 1: the
//      migration path moves 240 MB in about 70 ms each way on this system).  1: also prefetched to the host after a
//      synchronous launch (slower still).  2 / 3: as 1 / 0 without the preferred-location advice.
// OWL_MANAGED_POLICY selects another one (measurements, or a host program that relies on ONE address for both sides).
int managed_policy() {
  static const int p = [] {
    const char *e = getenv("OWL_MANAGED_POLICY");
    return e ? atoi(e) : 4;
  }();
  return p;
}

const void *Buffer::host_pointer() {
  if (!mirrored) return ptr;
  if (device_newer && ptr) {
    OWL_HIP(hipDeviceSynchronize());  // launches of every OWLParams stream that may still write the buffer
    OWL_HIP(hipMemcpy(mirror, ptr, bytes() ? bytes() : 16, hipMemcpyDeviceToHost));
    device_newer = false;
  }
  handed_out = true;  // the caller holds a writable pointer from now on
  return mirror;
}

// Before a launch: what the host may have written through the pointer it holds.  Not while the device side is newer
// than the mirror (an asynchronous launch has run and no synchronisation point has passed): the host has had no
// legal look at the buffer since, and the copy would overwrite that launch's results.
void Buffer::write_back(hipStream_t s) {
  if (!mirrored || !handed_out || device_newer || !ptr) return;
  if (managed_policy() == 5) return;  // the application has declared that its host code only READS managed buffers
  OWL_HIP(hipMemcpyAsync(ptr, mirror, bytes() ? bytes() : 16, hipMemcpyHostToDevice, s));
}

// At a synchronisation point: every mirror whose pointer the host holds shows what the launches so far have written.
static void refresh_mirrors(Context &c) {
  bool synced = false;
  for (auto &wb : c.buffers)
    if (auto b = wb.lock())
      if (b->mirrored && b->handed_out && b->device_newer && b->ptr) {
        if (!synced) OWL_HIP(hipDeviceSynchronize());  // launches of every OWLParams stream that may still write the buffer
        synced = true;
        OWL_HIP(hipMemcpy(b->mirror, b->ptr, b->bytes() ? b->bytes() : 16, hipMemcpyDeviceToHost));
        b->device_newer = false;
      }
}

bool launch_order_enabled() {
  const char *e = getenv("OWL_LAUNCH_ORDER");
  return !(e && atoi(e) == 0);
}

void launch(RayGen &rg, int dx, int dy, Params *lp, bool sync) {
  Context &c = *rg.ctx;
  if (!rg.kernel) fail("raygen program not built: call owlBuildPrograms / owlBuildPipeline / owlBuildSBT before launching");
  if (!rg.data.ptr) fail("shader binding table not built: call owlBuildSBT before launching");
  if (dx < 0 || dy < 0) fail("negative launch dimensions");
  hipStream_t s = lp ? lp->stream : c.stream;
  if (lp) {
    Module &m = *rg.module;
    if (!m.params_ptr) fail("launch params given, but the raygen's module defines no `optixLaunchParams`");
    OWL_HIP(hipStreamSynchronize(s));  // the previous launch on this stream may still read `staging`
    std::vector<uint8_t> &blob = lp->staging;
    lp->materialise(blob);
    if (blob.size() > m.params_bytes) fail("launch params struct is larger than the module's optixLaunchParams");
    if (m.params_in_use && m.params_stream != s) OWL_HIP(hipStreamWaitEvent(s, m.params_free, 0));
    OWL_HIP(hipMemcpyAsync(m.params_ptr, blob.data(), blob.size(), hipMemcpyHostToDevice, s));
  }
  // managed buffers may have been paged to the host by the application between launches
  for (auto &wb : c.buffers)
    if (auto b = wb.lock())
      if (b->bkind == BufferKind::Managed && b->ptr) {
        if (b->mirrored) {
          b->write_back(s);       // what the host may have written through owlBufferGetPointer's pointer
          b->device_newer = true;  // ... and the launch may write the device side
        } else if (hipMemPrefetchAsync(b->ptr, b->bytes() ? b->bytes() : 16, c.device, s) != hipSuccess) {
          (void)hipGetLastError();
        }
      }
  rec::LaunchDesc desc;
  desc.dims[0] = (uint32_t)dx;
  desc.dims[1] = (uint32_t)dy;
  desc.dims[2] = 1;
  desc.num_miss = (uint32_t)c.miss_by_ray_type.size();
  desc.raygen_data = rg.data.ptr;
  desc.miss = (const rec::MissRecord *)c.miss_records.ptr;
  // The order of a launch's indices is the backend's to choose (LaunchDesc::order): a 1-D launch with as many indices as
  // the context's one built user geometry group has primitives runs them in that group's Morton order -- in the
  // neighbour-query programs index i is the query AT primitive i, and 64 neighbouring queries walk the same nodes
  // (the reference's unchanged sample, 1 M points: 8.8 -> 8.0 ms per launch; the rest is the sample's own intersection
  // program keeping its k best by insertion into 24-byte records in global memory).  OWL_LAUNCH_ORDER=0: index = thread.
  desc.order = nullptr;
  if (dy == 1 && dx > 1 && launch_order_enabled()) {
    const UserGeomGroup *only = nullptr;
    int built_groups = 0;
    for (auto &wg : c.user_groups)
      if (auto g = wg.lock())
        if (g->built) {
          built_groups++;
          only = g.get();
        }
    if (built_groups == 1 && only->bvh.built() && only->bvh.size() == (int64_t)dx) desc.order = only->bvh.view().prim_id;
  }
  void *args[] = {(void *)&desc};
  launch_kernel(rg.kernel, (uint64_t)dx * (uint64_t)dy, rec::kRaygenBlock, args, s);
  if (lp) {
    Module &m = *rg.module;
    if (!m.params_free) OWL_HIP(hipEventCreateWithFlags(&m.params_free, hipEventDisableTiming));
    OWL_HIP(hipEventRecord(m.params_free, s));
    m.params_stream = s;
    m.params_in_use = true;
  }
  if (sync && (managed_policy() == 1 || managed_policy() == 2)) {
    // owlLaunch2D returns to host code that is about to read its managed buffers: one bulk transfer instead of a
    // page fault per 4 KB touched
    for (auto &wb : c.buffers)
      if (auto b = wb.lock())
        if (b->bkind == BufferKind::Managed && b->ptr && !b->mirrored)
          if (hipMemPrefetchAsync(b->ptr, b->bytes() ? b->bytes() : 16, hipCpuDeviceId, s) != hipSuccess) (void)hipGetLastError();
  }
  if (sync) {
    OWL_HIP(hipStreamSynchronize(s));
    refresh_mirrors(c);
  }
}

[[noreturn]] void unsupported(const char *what) {
  fail(std::string(what) + " is not supported on this backend (outside the custom-primitive neighbour-query path)");
}

// ---- typed variable setters -------------------------------------------------------------------
template <typename T>
struct TypeTag;
#define OWL_TAG(T, base)                             \
  template <>                                        \
  struct TypeTag<T> {                                \
    static constexpr int first = (int)base;          \
  };
OWL_TAG(bool, OWL_BOOL)
OWL_TAG(int8_t, OWL_CHAR)
OWL_TAG(uint8_t, OWL_UCHAR)
OWL_TAG(int16_t, OWL_SHORT)
OWL_TAG(uint16_t, OWL_USHORT)
OWL_TAG(float, OWL_FLOAT)
OWL_TAG(int32_t, OWL_INT)
OWL_TAG(uint32_t, OWL_UINT)
OWL_TAG(double, OWL_DOUBLE)
OWL_TAG(int64_t, OWL_LONG)
OWL_TAG(uint64_t, OWL_ULONG)

template <typename T, int N>
void set_values(SBTObject &o, int index, const T *v) {
  const VarDecl &d = (*o.decls)[index];
  if ((int)d.type != TypeTag<T>::first + (N - 1))
    fail("variable '" + d.name + "' was declared with OWLDataType " + std::to_string((int)d.type) +
         " but set with a value of type " + std::to_string(TypeTag<T>::first + N - 1) + " (type mismatch)");
  std::memcpy(o.host.data() + d.offset, v, sizeof(T) * N);
}

void set_buffer(SBTObject &o, int index, const void *buffer_handle) {
  const VarDecl &d = (*o.decls)[index];
  if (d.type != OWL_BUFFER_POINTER && d.type != OWL_BUFFER && d.type != OWL_BUFFER_SIZE)
    fail("variable '" + d.name + "' is not a buffer variable");
  if (!buffer_handle)
    o.buffer_refs.erase(index);
  else
    o.buffer_refs[index] = get<Buffer>(buffer_handle, Kind::Buffer);
}
void set_group(SBTObject &o, int index, const void *group_handle) {
  const VarDecl &d = (*o.decls)[index];
  if (d.type != OWL_GROUP) fail("variable '" + d.name + "' is not of type OWL_GROUP");
  if (!group_handle) {
    o.group_refs.erase(index);
    return;
  }
  auto g = get<Group>(group_handle, Kind::Group);
  // reference: only instance groups can be traced (Variable.cpp:336-341)
  if (!g->is_instance) fail("only instance groups may be assigned to an OWL_GROUP variable (wrap the geometry group in owlInstanceGroupCreate)");
  o.group_refs[index] = g;
}
void set_raw(SBTObject &o, int index, const void *src) {
  const VarDecl &d = (*o.decls)[index];
  if (!src) fail("owl*SetRaw: NULL value pointer");
  std::memcpy(o.host.data() + d.offset, src, size_of_type(d.type));
}
void set_pointer(SBTObject &o, int index, const void *value) {
  const VarDecl &d = (*o.decls)[index];
  if (d.type != OWL_RAW_POINTER) fail("variable '" + d.name + "' is not of type OWL_RAW_POINTER");
  std::memcpy(o.host.data() + d.offset, &value, 8);
}

}  // namespace

// =================================================================================================
// C-ABI
// =================================================================================================
#define CTX(h) get<Context>(h, Kind::Context)

OWL_API OWLContext owlContextCreate(int32_t *requestedDeviceIDs, int numDevices) {
  int visible = 0;
  if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0)
    fail("no HIP device visible; this OWL backend has no CPU fallback");
  int dev = 0;
  if (requestedDeviceIDs && numDevices > 0) dev = requestedDeviceIDs[0];
  if (numDevices > 1)
    std::fprintf(stderr, "#owl(mi355x): context spans one device; use one process per GPU for more (requested %d)\n", numDevices);
  if (dev < 0 || dev >= visible) fail("requested device id out of range");
  auto c = std::make_shared<Context>();
  c->device = dev;
  OWL_HIP(hipSetDevice(dev));
  OWL_HIP(hipStreamCreate(&c->stream));
  Handle *h = new Handle{c, c.get()};
  return (OWLContext)h;
}

// OWL_MI355X_DUMP_BUFFERS=<directory> (tests, and a maintainer comparing runs: INTEGRATION.md): when a context is destroyed,
// the final contents of its MANAGED buffers -- in the unchanged samples: the frameBuffer of Neigh records, which no API call
// exports and whose dump loop the reference has commented out (hostCode.cpp:312-321) -- go to
// <directory>/managed_<creation index>_<bytes>.bin.
static void dump_managed_buffers(const std::set<Handle *> &all) {
  const char *dir = getenv("OWL_MI355X_DUMP_BUFFERS");
  if (!dir || !*dir) return;
  for (Handle *h : all) {
    if (!h || !h->obj || h->obj->kind != Kind::Buffer) continue;
    auto b = std::static_pointer_cast<Buffer>(h->obj);
    if (b->bkind != BufferKind::Managed || !b->ptr || !b->bytes()) continue;
    std::vector<uint8_t> host(b->bytes());
    // (mirror form: the host's copy is the newer one if it holds the pointer and no launch has run since its last refresh)
    const void *src = (b->mirror && b->handed_out && !b->device_newer) ? b->mirror : b->ptr;
    if (hipMemcpy(host.data(), src, b->bytes(), hipMemcpyDefault) != hipSuccess) continue;
    const std::string path = std::string(dir) + "/managed_" + std::to_string(b->serial) + "_" + std::to_string(b->bytes()) + ".bin";
    if (FILE *f = std::fopen(path.c_str(), "wb")) {
      std::fwrite(host.data(), 1, host.size(), f);
      std::fclose(f);
    }
  }
}

OWL_API void owlContextDestroy(OWLContext context) {
  auto c = CTX(context);
  (void)hipDeviceSynchronize();
  std::set<Handle *> all;
  {
    std::lock_guard<std::mutex> g(c->mtx);
    all.swap(c->handles);
  }
  dump_managed_buffers(all);
  for (Handle *h : all) delete h;
  c->miss_by_ray_type.clear();
  if (c->stream) (void)hipStreamDestroy(c->stream);
  c->stream = nullptr;
  delete (Handle *)context;
}

OWL_API int32_t owlGetDeviceCount(OWLContext context) {
  CTX(context);
  return 1;
}
OWL_API void owlEnableMotionBlur(OWLContext) { unsupported("motion blur"); }
OWL_API void owlContextSetRayTypeCount(OWLContext context, size_t numRayTypes) {
  auto c = CTX(context);
  if (numRayTypes < 1 || numRayTypes > (size_t)rec::kMaxRayTypes) fail("ray type count must be 1.." + std::to_string(rec::kMaxRayTypes));
  c->num_ray_types = numRayTypes;
}
OWL_API void owlSetMaxInstancingDepth(OWLContext context, int32_t depth) {
  CTX(context);
  if (depth > 1) unsupported("instancing depth > 1");
}
OWL_API CUstream owlContextGetStream(OWLContext context, int) { return CTX(context)->stream; }
OWL_API OptixDeviceContext owlContextGetOptixContext(OWLContext, int) { unsupported("owlContextGetOptixContext (there is no OptiX)"); }

OWL_API OWLModule owlModuleCreate(OWLContext context, const char *code) {
  auto c = CTX(context);
  auto m = std::make_shared<Module>(c.get(), code);
  c->modules.push_back(m);
  return (OWLModule)c->make_handle(m);
}

OWL_API void owlBuildPrograms(OWLContext context) { build_programs(*CTX(context)); }
OWL_API void owlBuildPipeline(OWLContext context) {
  auto c = CTX(context);
  if (!c->programs_built) build_programs(*c);  // nothing else to link: programs are symbols of loaded code objects
}
OWL_API void owlBuildSBT(OWLContext context, OWLBuildSBTFlags flags) {
  auto c = CTX(context);
  if (!c->programs_built) build_programs(*c);
  build_sbt(*c, (int)flags);
}

OWL_API OWLGeomType owlGeomTypeCreate(OWLContext context, OWLGeomKind kind, size_t bytes, OWLVarDecl *vars, int numVars) {
  auto c = CTX(context);
  if (kind != OWL_GEOMETRY_USER) unsupported("geometry kinds other than OWL_GEOMETRY_USER");
  auto decls = copy_decls(vars, numVars);
  check_decls(decls, bytes);
  auto t = std::make_shared<GeomType>(c.get(), kind, bytes, std::move(decls));
  c->geom_types.push_back(t);
  return (OWLGeomType)c->make_handle(t);
}

static void set_prog(ProgRef *table, int rayType, OWLModule module, const char *name, Context &c) {
  if (rayType < 0 || rayType >= rec::kMaxRayTypes) fail("ray type out of range");
  if (!name) fail("program name is NULL");
  table[rayType].module = get<Module>(module, Kind::Module);
  table[rayType].name = name;
  table[rayType].fn = 0;
  c.programs_built = false;
}
OWL_API void owlGeomTypeSetIntersectProg(OWLGeomType type, int rayType, OWLModule module, const char *name) {
  auto t = get<GeomType>(type, Kind::GeomType);
  set_prog(t->intersect, rayType, module, name, *t->ctx);
}
OWL_API void owlGeomTypeSetClosestHit(OWLGeomType type, int rayType, OWLModule module, const char *name) {
  auto t = get<GeomType>(type, Kind::GeomType);
  set_prog(t->closest_hit, rayType, module, name, *t->ctx);
}
OWL_API void owlGeomTypeSetAnyHit(OWLGeomType type, int rayType, OWLModule module, const char *name) {
  auto t = get<GeomType>(type, Kind::GeomType);
  set_prog(t->any_hit, rayType, module, name, *t->ctx);
}
OWL_API void owlGeomTypeSetBoundsProg(OWLGeomType type, OWLModule module, const char *name) {
  auto t = get<GeomType>(type, Kind::GeomType);
  if (!name) fail("program name is NULL");
  t->bounds.module = get<Module>(module, Kind::Module);
  t->bounds.name = name;
  t->bounds_kernel = nullptr;
  t->ctx->programs_built = false;
}

OWL_API OWLGeom owlGeomCreate(OWLContext context, OWLGeomType type) {
  auto c = CTX(context);
  auto g = std::make_shared<Geom>(c.get(), get<GeomType>(type, Kind::GeomType));
  c->geoms.push_back(g);
  return (OWLGeom)c->make_handle(g);
}
OWL_API void owlGeomSetPrimCount(OWLGeom geom, size_t n) { get<Geom>(geom, Kind::Geom)->prim_count = n; }

OWL_API OWLParams owlParamsCreate(OWLContext context, size_t bytes, OWLVarDecl *vars, int numVars) {
  auto c = CTX(context);
  auto decls = copy_decls(vars, numVars);
  check_decls(decls, bytes);
  auto p = std::make_shared<Params>(c.get(), bytes, std::move(decls));
  return (OWLParams)c->make_handle(p);
}
OWL_API OWLRayGen owlRayGenCreate(OWLContext context, OWLModule module, const char *name, size_t bytes, OWLVarDecl *vars, int numVars) {
  auto c = CTX(context);
  if (!name) fail("program name is NULL");
  auto decls = copy_decls(vars, numVars);
  check_decls(decls, bytes);
  auto r = std::make_shared<RayGen>(c.get(), get<Module>(module, Kind::Module), name, bytes, std::move(decls));
  c->raygens.push_back(r);
  c->programs_built = false;
  return (OWLRayGen)c->make_handle(r);
}
OWL_API OWLMissProg owlMissProgCreate(OWLContext context, OWLModule module, const char *name, size_t bytes, OWLVarDecl *vars, int numVars) {
  auto c = CTX(context);
  if (!name) fail("program name is NULL");
  auto decls = copy_decls(vars, numVars);
  check_decls(decls, bytes);
  auto m = std::make_shared<MissProg>(c.get(), get<Module>(module, Kind::Module), name, bytes, std::move(decls));
  c->miss_progs.push_back(m);
  // the i-th created miss program serves ray type i unless owlMissProgSet says otherwise (owl_host.h:444-449)
  if (c->miss_by_ray_type.size() < (size_t)rec::kMaxRayTypes) c->miss_by_ray_type.push_back(m);
  c->programs_built = false;
  return (OWLMissProg)c->make_handle(m);
}
OWL_API void owlMissProgSet(OWLContext context, int rayType, OWLMissProg miss) {
  auto c = CTX(context);
  if (rayType < 0 || rayType >= rec::kMaxRayTypes) fail("ray type out of range");
  if (c->miss_by_ray_type.size() <= (size_t)rayType) c->miss_by_ray_type.resize(rayType + 1);
  c->miss_by_ray_type[rayType] = get<MissProg>(miss, Kind::MissProg);
}

OWL_API OWLGroup owlUserGeomGroupCreate(OWLContext context, size_t n, OWLGeom *geoms) {
  auto c = CTX(context);
  if (n == 0 || !geoms) fail("owlUserGeomGroupCreate needs at least one geometry");
  OWL_HIP(hipSetDevice(c->device));
  std::vector<std::shared_ptr<Geom>> list;
  for (size_t i = 0; i < n; i++) list.push_back(get<Geom>(geoms[i], Kind::Geom));
  auto g = std::make_shared<UserGeomGroup>(c.get(), std::move(list));
  c->user_groups.push_back(g);
  return (OWLGroup)c->make_handle(g);
}
OWL_API OWLGroup owlTrianglesGeomGroupCreate(OWLContext, size_t, OWLGeom *) { unsupported("triangle geometry groups"); }
OWL_API OWLGroup owlInstanceGroupCreate(OWLContext context, size_t n, const OWLGroup *groups, const uint32_t *ids,
                                        const float *xfms, OWLMatrixFormat fmt) {
  auto c = CTX(context);
  OWL_HIP(hipSetDevice(c->device));
  auto g = std::make_shared<InstanceGroup>(c.get(), n);
  for (size_t i = 0; i < n; i++) {
    if (groups && groups[i]) g->children[i] = get<Group>(groups[i], Kind::Group);
    if (ids) g->ids[i] = ids[i];
    if (xfms) g->set_transform(i, xfms + 12 * i, fmt);
  }
  return (OWLGroup)c->make_handle(g);
}
OWL_API void owlInstanceGroupSetChild(OWLGroup group, int which, OWLGroup child) {
  auto g = get<Group>(group, Kind::Group);
  if (!g->is_instance) fail("not an instance group");
  auto ig = std::static_pointer_cast<InstanceGroup>(g);
  if (which < 0 || (size_t)which >= ig->children.size()) fail("instance index out of range");
  ig->children[which] = get<Group>(child, Kind::Group);
}
OWL_API void owlInstanceGroupSetTransform(OWLGroup group, int which, const float *floats, OWLMatrixFormat fmt) {
  auto g = get<Group>(group, Kind::Group);
  if (!g->is_instance) fail("not an instance group");
  if (!floats) fail("transform pointer is NULL");
  std::static_pointer_cast<InstanceGroup>(g)->set_transform((size_t)which, floats, fmt);
}
OWL_API void owlInstanceGroupSetTransforms(OWLGroup, uint32_t, const float *, OWLMatrixFormat) { unsupported("motion-blur transforms"); }
OWL_API void owlInstanceGroupSetInstanceIDs(OWLGroup group, const uint32_t *ids) {
  auto g = get<Group>(group, Kind::Group);
  if (!g->is_instance) fail("not an instance group");
  auto ig = std::static_pointer_cast<InstanceGroup>(g);
  for (size_t i = 0; i < ig->ids.size(); i++) ig->ids[i] = ids ? ids[i] : (uint32_t)i;
}
OWL_API void owlGroupBuildAccel(OWLGroup group) {
  auto g = get<Group>(group, Kind::Group);
  OWL_HIP(hipSetDevice(g->ctx->device));
  g->build(false);
}
OWL_API void owlGroupRefitAccel(OWLGroup group) {
  auto g = get<Group>(group, Kind::Group);
  OWL_HIP(hipSetDevice(g->ctx->device));
  g->build(true);
}
OWL_API OptixTraversableHandle owlGroupGetTraversable(OWLGroup group, int) { return get<Group>(group, Kind::Group)->traversable(); }

static OWLBuffer make_buffer(OWLContext context, BufferKind k, OWLDataType type, size_t count, const void *init) {
  auto c = CTX(context);
  OWL_HIP(hipSetDevice(c->device));
  auto b = std::make_shared<Buffer>(c.get(), k, type, count);
  b->allocate(init);
  c->buffers.push_back(b);
  prune(c->buffers);
  return (OWLBuffer)c->make_handle(b);
}
OWL_API OWLBuffer owlDeviceBufferCreate(OWLContext c, OWLDataType t, size_t n, const void *init) { return make_buffer(c, BufferKind::Device, t, n, init); }
OWL_API OWLBuffer owlManagedMemoryBufferCreate(OWLContext c, OWLDataType t, size_t n, const void *init) { return make_buffer(c, BufferKind::Managed, t, n, init); }
OWL_API OWLBuffer owlHostPinnedBufferCreate(OWLContext c, OWLDataType t, size_t n) { return make_buffer(c, BufferKind::HostPinned, t, n, nullptr); }
OWL_API OWLBuffer owlGraphicsBufferCreate(OWLContext, OWLDataType, size_t, cudaGraphicsResource_t) { unsupported("graphics-interop buffers"); }
OWL_API void owlGraphicsBufferMap(OWLBuffer) { unsupported("graphics-interop buffers"); }
OWL_API void owlGraphicsBufferUnmap(OWLBuffer) { unsupported("graphics-interop buffers"); }
OWL_API const void *owlBufferGetPointer(OWLBuffer buffer, int) { return get<Buffer>(buffer, Kind::Buffer)->host_pointer(); }
OWL_API void owlBufferResize(OWLBuffer buffer, size_t n) { get<Buffer>(buffer, Kind::Buffer)->resize(n); }
OWL_API void owlBufferDestroy(OWLBuffer buffer) {
  auto b = get<Buffer>(buffer, Kind::Buffer);
  b->release();
  b->count = 0;
  b->ctx->drop_handle((Handle *)buffer);
}
OWL_API void owlBufferUpload(OWLBuffer buffer, const void *host, size_t offset, size_t numBytes) {
  auto b = get<Buffer>(buffer, Kind::Buffer);
  if (!host) fail("owlBufferUpload: NULL source");
  if (numBytes == (size_t)-1) numBytes = b->bytes() - offset;
  if (offset + numBytes > b->bytes()) fail("owlBufferUpload: range exceeds the buffer");
  if (b->bkind == BufferKind::Device) {
    OWL_HIP(hipMemcpy((char *)b->ptr + offset, host, numBytes, hipMemcpyHostToDevice));
  } else if (b->mirrored) {
    std::memcpy((char *)b->host_pointer() + offset, host, numBytes);  // (refreshes the mirror first; written back before the next launch)
    if (managed_policy() == 5) OWL_HIP(hipMemcpy((char *)b->ptr + offset, host, numBytes, hipMemcpyHostToDevice));  // (no write-back under this policy)
  } else {
    std::memcpy((char *)b->ptr + offset, host, numBytes);
  }
}

OWL_API OWLTexture owlTexture2DCreate(OWLContext, OWLTexelFormat, uint32_t, uint32_t, const void *, OWLTextureFilterMode,
                                      OWLTextureAddressMode, OWLTextureColorSpace, uint32_t) { unsupported("textures"); }
OWL_API CUtexObject owlTextureGetObject(OWLTexture, int) { unsupported("textures"); }
OWL_API void owlTexture2DDestroy(OWLTexture) { unsupported("textures"); }
OWL_API void owlTrianglesSetVertices(OWLGeom, OWLBuffer, size_t, size_t, size_t) { unsupported("triangle meshes"); }
OWL_API void owlTrianglesSetMotionVertices(OWLGeom, size_t, OWLBuffer *, size_t, size_t, size_t) { unsupported("triangle meshes"); }
OWL_API void owlTrianglesSetIndices(OWLGeom, OWLBuffer, size_t, size_t, size_t) { unsupported("triangle meshes"); }

// synchronous like the reference's (RayGen.cpp:134-138: launchAsync on the dummy launch params, then their sync)
OWL_API void owlRayGenLaunch2D(OWLRayGen rayGen, int dx, int dy) { launch(*get<RayGen>(rayGen, Kind::RayGen), dx, dy, nullptr, true); }
OWL_API void owlLaunch2D(OWLRayGen rayGen, int dx, int dy, OWLParams params) {
  auto lp = get<Params>(params, Kind::Params);
  launch(*get<RayGen>(rayGen, Kind::RayGen), dx, dy, lp.get(), true);
}
OWL_API void owlAsyncLaunch2D(OWLRayGen rayGen, int dx, int dy, OWLParams params) {
  auto lp = get<Params>(params, Kind::Params);
  launch(*get<RayGen>(rayGen, Kind::RayGen), dx, dy, lp.get(), false);
}
OWL_API CUstream owlParamsGetCudaStream(OWLParams params, int) { return get<Params>(params, Kind::Params)->stream; }
OWL_API void owlLaunchSync(OWLParams params) {
  auto lp = get<Params>(params, Kind::Params);
  OWL_HIP(hipStreamSynchronize(lp->stream));
  refresh_mirrors(*lp->ctx);
}

// ---- releases, variable handles ------------------------------------------------------------------
static void release_handle(const void *h) {
  if (!h) return;
  Handle *hh = (Handle *)h;
  hh->ctx->drop_handle(hh);
}
OWL_API void owlGeomRelease(OWLGeom h) { get<Geom>(h, Kind::Geom); release_handle(h); }
OWL_API void owlVariableRelease(OWLVariable h) { get<Variable>(h, Kind::Variable); release_handle(h); }
OWL_API void owlModuleRelease(OWLModule h) { get<Module>(h, Kind::Module); release_handle(h); }
OWL_API void owlBufferRelease(OWLBuffer h) { get<Buffer>(h, Kind::Buffer); release_handle(h); }
OWL_API void owlRayGenRelease(OWLRayGen h) { get<RayGen>(h, Kind::RayGen); release_handle(h); }
OWL_API void owlGroupRelease(OWLGroup h) { get<Group>(h, Kind::Group); release_handle(h); }

static OWLVariable get_variable(const void *handle, Kind k, const char *name) {
  auto o = get_sbt(handle, k);
  const int idx = o->find(name);
  auto v = std::make_shared<Variable>(o->ctx, o, idx);
  return (OWLVariable)o->ctx->make_handle(v);
}
OWL_API OWLVariable owlGeomGetVariable(OWLGeom h, const char *n) { return get_variable(h, Kind::Geom, n); }
OWL_API OWLVariable owlRayGenGetVariable(OWLRayGen h, const char *n) { return get_variable(h, Kind::RayGen, n); }
OWL_API OWLVariable owlMissProgGetVariable(OWLMissProg h, const char *n) { return get_variable(h, Kind::MissProg, n); }
OWL_API OWLVariable owlParamsGetVariable(OWLParams h, const char *n) { return get_variable(h, Kind::Params, n); }

// ---- generated setters -----------------------------------------------------------------------------
#define OWL_DEFINE_VARIABLE_SETTERS(sfx, T)                                                              \
  OWL_API void owlVariableSet1##sfx(OWLVariable var, T v) {                                              \
    auto x = get<Variable>(var, Kind::Variable);                                                         \
    set_values<T, 1>(*x->owner, x->index, &v);                                                           \
  }                                                                                                      \
  OWL_API void owlVariableSet2##sfx(OWLVariable var, T a, T b) {                                         \
    auto x = get<Variable>(var, Kind::Variable);                                                         \
    T v[2] = {a, b};                                                                                     \
    set_values<T, 2>(*x->owner, x->index, v);                                                            \
  }                                                                                                      \
  OWL_API void owlVariableSet3##sfx(OWLVariable var, T a, T b, T c) {                                    \
    auto x = get<Variable>(var, Kind::Variable);                                                         \
    T v[3] = {a, b, c};                                                                                  \
    set_values<T, 3>(*x->owner, x->index, v);                                                            \
  }                                                                                                      \
  OWL_API void owlVariableSet4##sfx(OWLVariable var, T a, T b, T c, T d) {                               \
    auto x = get<Variable>(var, Kind::Variable);                                                         \
    T v[4] = {a, b, c, d};                                                                               \
    set_values<T, 4>(*x->owner, x->index, v);                                                            \
  }                                                                                                      \
  OWL_API void owlVariableSet2##sfx##v(OWLVariable var, const T *v) {                                    \
    auto x = get<Variable>(var, Kind::Variable);                                                         \
    set_values<T, 2>(*x->owner, x->index, v);                                                            \
  }                                                                                                      \
  OWL_API void owlVariableSet3##sfx##v(OWLVariable var, const T *v) {                                    \
    auto x = get<Variable>(var, Kind::Variable);                                                         \
    set_values<T, 3>(*x->owner, x->index, v);                                                            \
  }                                                                                                      \
  OWL_API void owlVariableSet4##sfx##v(OWLVariable var, const T *v) {                                    \
    auto x = get<Variable>(var, Kind::Variable);                                                         \
    set_values<T, 4>(*x->owner, x->index, v);                                                            \
  }
OWL_FOREACH_SCALAR(OWL_DEFINE_VARIABLE_SETTERS)

#define OWL_DEFINE_OBJECT_SETTERS_(Obj, HandleT, KindV, sfx, T)                                          \
  OWL_API void owl##Obj##Set1##sfx(HandleT h, const char *name, T v) {                                   \
    auto o = get_sbt(h, KindV);                                                                          \
    set_values<T, 1>(*o, o->find(name), &v);                                                             \
  }                                                                                                      \
  OWL_API void owl##Obj##Set2##sfx(HandleT h, const char *name, T a, T b) {                              \
    auto o = get_sbt(h, KindV);                                                                          \
    T v[2] = {a, b};                                                                                     \
    set_values<T, 2>(*o, o->find(name), v);                                                              \
  }                                                                                                      \
  OWL_API void owl##Obj##Set3##sfx(HandleT h, const char *name, T a, T b, T c) {                         \
    auto o = get_sbt(h, KindV);                                                                          \
    T v[3] = {a, b, c};                                                                                  \
    set_values<T, 3>(*o, o->find(name), v);                                                              \
  }                                                                                                      \
  OWL_API void owl##Obj##Set4##sfx(HandleT h, const char *name, T a, T b, T c, T d) {                    \
    auto o = get_sbt(h, KindV);                                                                          \
    T v[4] = {a, b, c, d};                                                                               \
    set_values<T, 4>(*o, o->find(name), v);                                                              \
  }                                                                                                      \
  OWL_API void owl##Obj##Set2##sfx##v(HandleT h, const char *name, const T *v) {                         \
    auto o = get_sbt(h, KindV);                                                                          \
    set_values<T, 2>(*o, o->find(name), v);                                                              \
  }                                                                                                      \
  OWL_API void owl##Obj##Set3##sfx##v(HandleT h, const char *name, const T *v) {                         \
    auto o = get_sbt(h, KindV);                                                                          \
    set_values<T, 3>(*o, o->find(name), v);                                                              \
  }                                                                                                      \
  OWL_API void owl##Obj##Set4##sfx##v(HandleT h, const char *name, const T *v) {                         \
    auto o = get_sbt(h, KindV);                                                                          \
    set_values<T, 4>(*o, o->find(name), v);                                                              \
  }
#define OWL_DEFINE_OBJECT_SETTERS(sfx, T)                                \
  OWL_DEFINE_OBJECT_SETTERS_(RayGen, OWLRayGen, Kind::RayGen, sfx, T)    \
  OWL_DEFINE_OBJECT_SETTERS_(MissProg, OWLMissProg, Kind::MissProg, sfx, T) \
  OWL_DEFINE_OBJECT_SETTERS_(Geom, OWLGeom, Kind::Geom, sfx, T)          \
  OWL_DEFINE_OBJECT_SETTERS_(Params, OWLParams, Kind::Params, sfx, T)
OWL_FOREACH_SCALAR(OWL_DEFINE_OBJECT_SETTERS)

OWL_API void owlVariableSetGroup(OWLVariable var, OWLGroup v) {
  auto x = get<Variable>(var, Kind::Variable);
  set_group(*x->owner, x->index, v);
}
OWL_API void owlVariableSetTexture(OWLVariable, OWLTexture) { unsupported("textures"); }
OWL_API void owlVariableSetBuffer(OWLVariable var, OWLBuffer v) {
  auto x = get<Variable>(var, Kind::Variable);
  set_buffer(*x->owner, x->index, v);
}
OWL_API void owlVariableSetRaw(OWLVariable var, const void *v) {
  auto x = get<Variable>(var, Kind::Variable);
  set_raw(*x->owner, x->index, v);
}
OWL_API void owlVariableSetPointer(OWLVariable var, const void *v) {
  auto x = get<Variable>(var, Kind::Variable);
  set_pointer(*x->owner, x->index, v);
}

#define OWL_DEFINE_OBJECT_REF_SETTERS(Obj, HandleT, KindV)                                               \
  OWL_API void owl##Obj##SetTexture(HandleT, const char *, OWLTexture) { unsupported("textures"); }      \
  OWL_API void owl##Obj##SetPointer(HandleT h, const char *name, const void *v) {                        \
    auto o = get_sbt(h, KindV);                                                                          \
    set_pointer(*o, o->find(name), v);                                                                   \
  }                                                                                                      \
  OWL_API void owl##Obj##SetBuffer(HandleT h, const char *name, OWLBuffer v) {                           \
    auto o = get_sbt(h, KindV);                                                                          \
    set_buffer(*o, o->find(name), v);                                                                    \
  }                                                                                                      \
  OWL_API void owl##Obj##SetGroup(HandleT h, const char *name, OWLGroup v) {                             \
    auto o = get_sbt(h, KindV);                                                                          \
    set_group(*o, o->find(name), v);                                                                     \
  }                                                                                                      \
  OWL_API void owl##Obj##SetRaw(HandleT h, const char *name, const void *v) {                            \
    auto o = get_sbt(h, KindV);                                                                          \
    set_raw(*o, o->find(name), v);                                                                       \
  }
OWL_DEFINE_OBJECT_REF_SETTERS(RayGen, OWLRayGen, Kind::RayGen)
OWL_DEFINE_OBJECT_REF_SETTERS(Geom, OWLGeom, Kind::Geom)
OWL_DEFINE_OBJECT_REF_SETTERS(Params, OWLParams, Kind::Params)
OWL_DEFINE_OBJECT_REF_SETTERS(MissProg, OWLMissProg, Kind::MissProg)
