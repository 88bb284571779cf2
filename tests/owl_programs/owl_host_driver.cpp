// owl_host_driver.cpp -- C++ host programs the GPU tests run against libowl_mi355x.so through the
// OWL C-ABI (include/owl/owl_host.h).  Written for the tests (not the reference's hostCode.cpp):
// inputs and outputs are raw binary files so that pytest can compare with the CPU checker.
//
//   owl_host_driver knn    <module.hsaco> <points.f32> <n> <k> <radius> <out_fb.bin> [held]
//       TrueKNN through owl*: the round loop of samples/s01-trueknn/hostCode.cpp:285-340 driven
//       over a device-program module that exports the reference's program names ("Spheres",
//       "rayGen") and struct layouts (GeomTypes.h).  Writes the n*k 24-byte Neigh records and
//       prints "rounds=<r>".
//   owl_host_driver count  <module.hsaco> <points.f32> <n> <radius> <out.bin>
//       tests/owl_programs/radius_programs.cu: per-point neighbour count / nearest / IS calls, then
//       a first-hit pass (closest-hit + miss programs); two geometries in one group.
//   owl_host_driver api    <api_programs.hsaco> <out.bin>
//       tests/owl_programs/api_programs.cu: OWL_BUFFER / OWL_BUFFER_SIZE / OWL_DEVICE variables, host-pinned output,
//       owlBufferUpload / Resize / Destroy, any-hit, instance transforms and ids, two OWLParams launched asynchronously.
//   owl_host_driver errors <module.hsaco>
//       the error conventions of SURVEY.md section 8(b); prints one PASS/FAIL line per check.
#include <owl/owl.h>
#include <owl/helper/cuda.h>

#include <cmath>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <string>
#include <vector>

static std::vector<char> read_file(const char *path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) {
    std::fprintf(stderr, "cannot open %s\n", path);
    std::exit(2);
  }
  std::vector<char> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  d.push_back(0);
  return d;
}
static void write_file(const char *path, const void *p, size_t n) {
  std::ofstream f(path, std::ios::binary);
  f.write((const char *)p, (std::streamsize)n);
}

// layouts of samples/s01-trueknn/GeomTypes.h (restated: 24-byte Neigh, 12-byte Sphere, ...)
struct NeighRec {
  int ind;
  float dist;
  int numNeighbors;
  long long intersections;
};
struct Point3 {
  float x, y, z;
};
struct SpheresGeomRec {
  Point3 *prims;
  float rad;
};
struct GlobalsRec {
  NeighRec *frameBuffer;
  int k;
  Point3 *spheres;
  float distRadius;
};
struct RayGenRec {
  uint32_t *fbPtr;
  int fbSize[2];
  OptixTraversableHandle world;
  int sbtOffset;
  float camera[12];
};

static int run_knn(int argc, char **argv) {
  if (argc < 8) return 2;
  std::vector<char> code = read_file(argv[2]);
  const size_t n = (size_t)std::atoll(argv[4]);
  const int k = std::atoi(argv[5]);
  float radius = (float)std::atof(argv[6]);
  std::vector<char> raw = read_file(argv[3]);
  const Point3 *pts = (const Point3 *)raw.data();
  std::vector<NeighRec> init(n * (size_t)k);
  for (auto &e : init) {
    std::memset(&e, 0, sizeof e);
    e.ind = -1;
    e.dist = (float)3.402823466e+38;
    e.numNeighbors = k;
  }
  OWLContext ctx = owlContextCreate(nullptr, 1);
  OWLModule mod = owlModuleCreate(ctx, code.data());
  OWLVarDecl geomVars[] = {{"prims", OWL_BUFPTR, OWL_OFFSETOF(SpheresGeomRec, prims)},
                           {"rad", OWL_FLOAT, OWL_OFFSETOF(SpheresGeomRec, rad)},
                           {nullptr, OWL_INVALID_TYPE, 0}};
  OWLGeomType type = owlGeomTypeCreate(ctx, OWL_GEOMETRY_USER, sizeof(SpheresGeomRec), geomVars, -1);
  owlGeomTypeSetIntersectProg(type, 0, mod, "Spheres");
  owlGeomTypeSetBoundsProg(type, mod, "Spheres");
  owlBuildPrograms(ctx);
  OWLBuffer fb = owlManagedMemoryBufferCreate(ctx, OWL_USER_TYPE(NeighRec), init.size(), init.data());
  OWLBuffer points = owlDeviceBufferCreate(ctx, OWL_USER_TYPE(Point3), n, pts);
  OWLGeom geom = owlGeomCreate(ctx, type);
  owlGeomSetPrimCount(geom, n);
  owlGeomSetBuffer(geom, "prims", points);
  owlGeomSet1f(geom, "rad", radius);
  OWLVarDecl lpVars[] = {{"frameBuffer", OWL_BUFPTR, OWL_OFFSETOF(GlobalsRec, frameBuffer)},
                         {"k", OWL_INT, OWL_OFFSETOF(GlobalsRec, k)},
                         {"spheres", OWL_BUFPTR, OWL_OFFSETOF(GlobalsRec, spheres)},
                         {"distRadius", OWL_FLOAT, OWL_OFFSETOF(GlobalsRec, distRadius)},
                         {nullptr, OWL_INVALID_TYPE, 0}};
  OWLParams lp = owlParamsCreate(ctx, sizeof(GlobalsRec), lpVars, -1);
  owlParamsSetBuffer(lp, "frameBuffer", fb);
  owlParamsSet1i(lp, "k", k);
  owlParamsSetBuffer(lp, "spheres", points);
  owlParamsSet1f(lp, "distRadius", radius);
  OWLGroup blas = owlUserGeomGroupCreate(ctx, 1, &geom);
  owlGroupBuildAccel(blas);
  OWLGroup world = owlInstanceGroupCreate(ctx, 1, &blas);
  owlGroupBuildAccel(world);
  OWLVarDecl rgVars[] = {{"fbSize", OWL_INT2, OWL_OFFSETOF(RayGenRec, fbSize)},
                         {"world", OWL_GROUP, OWL_OFFSETOF(RayGenRec, world)},
                         {nullptr, OWL_INVALID_TYPE, 0}};
  OWLRayGen rg = owlRayGenCreate(ctx, mod, "rayGen", sizeof(RayGenRec), rgVars, -1);
  owlRayGenSet2i(rg, "fbSize", (int)n, 1);
  owlRayGenSetGroup(rg, "world", world);
  owlBuildPrograms(ctx);
  owlBuildPipeline(ctx);
  owlBuildSBT(ctx);
  // "held" (9th argument): the pointer of the managed frameBuffer is fetched ONCE, before the first launch, and re-read
  // after every launch -- legal for managed memory in the reference API (one address, coherent after a synchronising
  // launch); a marker the host writes through it between two launches must reach the device and come back.
  const bool held = argc > 8 && !std::strcmp(argv[8], "held");
  NeighRec *held_rows = held ? (NeighRec *)owlBufferGetPointer(fb, 0) : nullptr;
  bool marker_ok = true;
  int rounds = 0;
  for (;;) {
    rounds++;
    if (held && rounds == 2 && k > 1) held_rows[1].intersections = 0x5eed5eedLL;  // (slot 1's counter: device code never touches it)
    owlLaunch2D(rg, (int)n, 1, lp);
    if (held && rounds == 2 && k > 1) {
      marker_ok = held_rows[1].intersections == 0x5eed5eedLL;
      held_rows[1].intersections = 0;
    }
    const NeighRec *rows = held ? held_rows : (const NeighRec *)owlBufferGetPointer(fb, 0);
    bool again = false;
    for (size_t j = 0; j < n; j++)
      if (rows[j * k].numNeighbors > 0) {
        again = true;
        break;
      }
    if (!again || rounds >= 64) break;
    radius *= 2;
    owlGeomSet1f(geom, "rad", radius);
    owlParamsSet1f(lp, "distRadius", radius);
    owlGroupRefitAccel(blas);
    owlGroupRefitAccel(world);
  }
  write_file(argv[7], held ? (const void *)held_rows : owlBufferGetPointer(fb, 0), init.size() * sizeof(NeighRec));
  std::printf("rounds=%d final_radius=%.9g%s\n", rounds, radius, held ? (marker_ok ? " held_marker=ok" : " held_marker=LOST") : "");
  owlContextDestroy(ctx);
  return 0;
}

// tests/owl_programs/radius_programs.cu layouts
struct BallsGeomRec {
  Point3 *centers;
  float radius;
};
struct CountParamsRec {
  int *count;
  float *nearest;
  long long *calls;
  Point3 *queries;
  int first_hit_mode;
  int *first_hit;
};
struct CountRayGenRec {
  OptixTraversableHandle world;
  int n_queries;
};

static int run_count(int argc, char **argv) {
  if (argc < 7) return 2;
  std::vector<char> code = read_file(argv[2]);
  const size_t n = (size_t)std::atoll(argv[4]);
  const float radius = (float)std::atof(argv[5]);
  std::vector<char> raw = read_file(argv[3]);
  const Point3 *pts = (const Point3 *)raw.data();
  OWLContext ctx = owlContextCreate(nullptr, 1);
  OWLModule mod = owlModuleCreate(ctx, code.data());
  OWLVarDecl geomVars[] = {{"centers", OWL_BUFPTR, OWL_OFFSETOF(BallsGeomRec, centers)},
                           {"radius", OWL_FLOAT, OWL_OFFSETOF(BallsGeomRec, radius)}};
  OWLGeomType type = owlGeomTypeCreate(ctx, OWL_GEOMETRY_USER, sizeof(BallsGeomRec), geomVars, 2);
  owlGeomTypeSetIntersectProg(type, 0, mod, "Balls");
  owlGeomTypeSetClosestHit(type, 0, mod, "Balls");
  owlGeomTypeSetBoundsProg(type, mod, "Balls");
  OWLVarDecl missVars[] = {{nullptr, OWL_INVALID_TYPE, 0}};
  owlMissProgCreate(ctx, mod, "nothing", 0, missVars, -1);
  owlBuildPrograms(ctx);
  // the point set is split over TWO geometries of one group: primitive ids are per geometry,
  // so the second buffer is a view (device copy) of the tail
  const size_t n0 = n / 3, n1 = n - n0;
  OWLBuffer all = owlDeviceBufferCreate(ctx, OWL_USER_TYPE(Point3), n, pts);
  OWLBuffer head = owlDeviceBufferCreate(ctx, OWL_USER_TYPE(Point3), n0, pts);
  OWLBuffer tail = owlDeviceBufferCreate(ctx, OWL_USER_TYPE(Point3), n1, pts + n0);
  OWLGeom g0 = owlGeomCreate(ctx, type), g1 = owlGeomCreate(ctx, type);
  owlGeomSetPrimCount(g0, n0);
  owlGeomSetBuffer(g0, "centers", head);
  owlGeomSet1f(g0, "radius", radius);
  owlGeomSetPrimCount(g1, n1);
  owlGeomSetBuffer(g1, "centers", tail);
  owlGeomSet1f(g1, "radius", radius);
  OWLGeom both[2] = {g0, g1};
  OWLGroup blas = owlUserGeomGroupCreate(ctx, 2, both);
  owlGroupBuildAccel(blas);
  OWLGroup world = owlInstanceGroupCreate(ctx, 1, &blas);
  owlGroupBuildAccel(world);
  std::vector<int> zeros(n, 0);
  std::vector<float> inf(n, INFINITY);
  std::vector<long long> zl(n, 0);
  OWLBuffer count = owlDeviceBufferCreate(ctx, OWL_INT, n, zeros.data());
  OWLBuffer nearest = owlDeviceBufferCreate(ctx, OWL_FLOAT, n, inf.data());
  OWLBuffer calls = owlDeviceBufferCreate(ctx, OWL_LONG, n, zl.data());
  OWLBuffer first = owlDeviceBufferCreate(ctx, OWL_INT, n, zeros.data());
  OWLVarDecl lpVars[] = {{"count", OWL_BUFPTR, OWL_OFFSETOF(CountParamsRec, count)},
                         {"nearest", OWL_BUFPTR, OWL_OFFSETOF(CountParamsRec, nearest)},
                         {"calls", OWL_BUFPTR, OWL_OFFSETOF(CountParamsRec, calls)},
                         {"queries", OWL_BUFPTR, OWL_OFFSETOF(CountParamsRec, queries)},
                         {"first_hit_mode", OWL_INT, OWL_OFFSETOF(CountParamsRec, first_hit_mode)},
                         {"first_hit", OWL_BUFPTR, OWL_OFFSETOF(CountParamsRec, first_hit)},
                         {nullptr, OWL_INVALID_TYPE, 0}};
  OWLParams lp = owlParamsCreate(ctx, sizeof(CountParamsRec), lpVars, -1);
  owlParamsSetBuffer(lp, "count", count);
  owlParamsSetBuffer(lp, "nearest", nearest);
  owlParamsSetBuffer(lp, "calls", calls);
  owlParamsSetBuffer(lp, "queries", all);
  owlParamsSetBuffer(lp, "first_hit", first);
  owlParamsSet1i(lp, "first_hit_mode", 0);
  OWLVarDecl rgVars[] = {{"world", OWL_GROUP, OWL_OFFSETOF(CountRayGenRec, world)},
                         {"n_queries", OWL_INT, OWL_OFFSETOF(CountRayGenRec, n_queries)},
                         {nullptr, OWL_INVALID_TYPE, 0}};
  OWLRayGen rg = owlRayGenCreate(ctx, mod, "queries", sizeof(CountRayGenRec), rgVars, -1);
  owlRayGenSetGroup(rg, "world", world);
  owlRayGenSet1i(rg, "n_queries", (int)n);
  owlBuildPrograms(ctx);
  owlBuildPipeline(ctx);
  owlBuildSBT(ctx);
  owlLaunch2D(rg, (int)n, 1, lp);
  owlParamsSet1i(lp, "first_hit_mode", 1);
  owlAsyncLaunch2D(rg, (int)n, 1, lp);
  owlLaunchSync(lp);
  // out file: count[n] i32 | nearest[n] f32 | calls[n] i64 | first_hit[n] i32
  std::vector<char> out(n * (4 + 4 + 8 + 4));
  char *p = out.data();
  CUDA_CHECK(cudaMemcpy(p, owlBufferGetPointer(count, 0), n * 4, cudaMemcpyDeviceToHost));
  p += n * 4;
  CUDA_CHECK(cudaMemcpy(p, owlBufferGetPointer(nearest, 0), n * 4, cudaMemcpyDeviceToHost));
  p += n * 4;
  CUDA_CHECK(cudaMemcpy(p, owlBufferGetPointer(calls, 0), n * 8, cudaMemcpyDeviceToHost));
  p += n * 8;
  CUDA_CHECK(cudaMemcpy(p, owlBufferGetPointer(first, 0), n * 4, cudaMemcpyDeviceToHost));
  write_file(argv[6], out.data(), out.size());
  std::printf("n0=%zu n1=%zu\n", n0, n1);
  owlContextDestroy(ctx);
  return 0;
}


// tests/owl_programs/api_programs.cu layouts
struct CubesGeomRec {
  Point3 *centers;
  float half;
  int reject_odd;
};
struct DeviceBufferRec {
  int type;
  int pad;
  unsigned long long count;
  void *data;
};
struct ApiParamsRec {
  DeviceBufferRec out;
  unsigned long long out_size;
  float *hit_t;
  Point3 *origins;
  int tag;
};
struct ApiRayGenRec {
  OptixTraversableHandle world;
  int device;
  int n;
};

// Scene: four cubes of half-width 0.25 on the line y = z = 0 at x = 1, 2, 3, 4 (primitives 0..3), instanced twice:
// instance 0 (id 70) as it is, instance 1 (id 71) moved by (+0.5, +10, 0).  Rays along +x.
//   pass A  reject_odd = 0: a ray from x = 0 on the line hits prim 0 of instance 0 at t = 0.75; one from y = 10 hits
//           prim 0 of instance 1 at t = 1.25 (object-space origin is shifted back by the instance transform).
//   pass B  reject_odd = 1 (any-hit ignores odd primitives): a ray starting at x = 1.5 skips prim 1 and ends on prim 2.
// Output file: for each pass and query int4 {prim, instance id, device, tag} + float t.
static int run_api(int argc, char **argv) {
  if (argc < 4) return 2;
  std::vector<char> code = read_file(argv[2]);
  int32_t two_ids[2] = {0, 0};
  OWLContext ctx = owlContextCreate(two_ids, 2);  // two devices asked for: the context spans one
  const int device_count = owlGetDeviceCount(ctx);
  OWLModule mod = owlModuleCreate(ctx, code.data());
  OWLVarDecl geomVars[] = {{"centers", OWL_BUFPTR, OWL_OFFSETOF(CubesGeomRec, centers)},
                           {"half", OWL_FLOAT, OWL_OFFSETOF(CubesGeomRec, half)},
                           {"reject_odd", OWL_INT, OWL_OFFSETOF(CubesGeomRec, reject_odd)},
                           {nullptr, OWL_INVALID_TYPE, 0}};
  OWLGeomType type = owlGeomTypeCreate(ctx, OWL_GEOMETRY_USER, sizeof(CubesGeomRec), geomVars, -1);
  owlGeomTypeSetIntersectProg(type, 0, mod, "Cubes");
  owlGeomTypeSetClosestHit(type, 0, mod, "Cubes");
  owlGeomTypeSetAnyHit(type, 0, mod, "Cubes");
  owlGeomTypeSetBoundsProg(type, mod, "Cubes");
  OWLVarDecl missVars[] = {{nullptr, OWL_INVALID_TYPE, 0}};
  owlMissProgCreate(ctx, mod, "none", 0, missVars, -1);
  owlBuildPrograms(ctx);
  // centers: created too small and with wrong contents, then resized and uploaded in two pieces
  Point3 wrong[2] = {{9, 9, 9}, {9, 9, 9}};
  OWLBuffer centers = owlDeviceBufferCreate(ctx, OWL_USER_TYPE(Point3), 2, wrong);
  owlBufferResize(centers, 4);
  Point3 cubes[4] = {{1, 0, 0}, {2, 0, 0}, {3, 0, 0}, {4, 0, 0}};
  owlBufferUpload(centers, cubes, 0, 2 * sizeof(Point3));
  owlBufferUpload(centers, cubes + 2, 2 * sizeof(Point3), 2 * sizeof(Point3));
  OWLBuffer scratch = owlDeviceBufferCreate(ctx, OWL_INT, 1000, nullptr);
  owlBufferDestroy(scratch);  // must not disturb anything else
  OWLGeom geom = owlGeomCreate(ctx, type);
  owlGeomSetPrimCount(geom, 4);
  owlGeomSetBuffer(geom, "centers", centers);
  owlGeomSet1f(geom, "half", 0.25f);
  owlGeomSet1i(geom, "reject_odd", 0);
  OWLGroup blas = owlUserGeomGroupCreate(ctx, 1, &geom);
  owlGroupBuildAccel(blas);
  OWLGroup both[2] = {blas, blas};
  uint32_t ids[2] = {70, 71};
  OWLGroup world = owlInstanceGroupCreate(ctx, 2, both, ids, nullptr, OWL_MATRIX_FORMAT_OWL);
  // OWL format: column major 4x3 = {vx, vy, vz, translation}
  const float moved[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0.5f, 10.f, 0.f};
  owlInstanceGroupSetTransform(world, 1, moved, OWL_MATRIX_FORMAT_OWL);
  owlGroupBuildAccel(world);
  const int n = 4;
  Point3 origins[n] = {{0, 0, 0}, {0, 10, 0}, {1.5f, 0.1f, 0}, {0, 5, 0}};
  OWLBuffer org = owlDeviceBufferCreate(ctx, OWL_USER_TYPE(Point3), n, origins);
  OWLBuffer outA = owlDeviceBufferCreate(ctx, OWL_INT4, n, nullptr), outB = owlDeviceBufferCreate(ctx, OWL_INT4, n, nullptr);
  OWLBuffer tA = owlHostPinnedBufferCreate(ctx, OWL_FLOAT, n), tB = owlHostPinnedBufferCreate(ctx, OWL_FLOAT, n);
  OWLVarDecl lpVars[] = {{"out", OWL_BUFFER, OWL_OFFSETOF(ApiParamsRec, out)},
                         {"out_size", OWL_BUFFER_SIZE, OWL_OFFSETOF(ApiParamsRec, out_size)},
                         {"hit_t", OWL_BUFPTR, OWL_OFFSETOF(ApiParamsRec, hit_t)},
                         {"origins", OWL_BUFPTR, OWL_OFFSETOF(ApiParamsRec, origins)},
                         {"tag", OWL_INT, OWL_OFFSETOF(ApiParamsRec, tag)},
                         {nullptr, OWL_INVALID_TYPE, 0}};
  OWLParams lpA = owlParamsCreate(ctx, sizeof(ApiParamsRec), lpVars, -1), lpB = owlParamsCreate(ctx, sizeof(ApiParamsRec), lpVars, -1);
  owlParamsSetBuffer(lpA, "out", outA);
  owlParamsSetBuffer(lpA, "out_size", outA);
  owlParamsSetBuffer(lpA, "hit_t", tA);
  owlParamsSetBuffer(lpA, "origins", org);
  owlParamsSet1i(lpA, "tag", 1001);
  owlParamsSetBuffer(lpB, "out", outB);
  owlParamsSetBuffer(lpB, "out_size", outB);
  owlParamsSetBuffer(lpB, "hit_t", tB);
  owlParamsSetBuffer(lpB, "origins", org);
  owlParamsSet1i(lpB, "tag", 2002);
  OWLVarDecl rgVars[] = {{"world", OWL_GROUP, OWL_OFFSETOF(ApiRayGenRec, world)},
                         {"device", OWL_DEVICE, OWL_OFFSETOF(ApiRayGenRec, device)},
                         {"n", OWL_INT, OWL_OFFSETOF(ApiRayGenRec, n)},
                         {nullptr, OWL_INVALID_TYPE, 0}};
  OWLRayGen rg = owlRayGenCreate(ctx, mod, "shoot", sizeof(ApiRayGenRec), rgVars, -1);
  owlRayGenSetGroup(rg, "world", world);
  owlRayGenSet1i(rg, "n", n);
  owlBuildPrograms(ctx);
  owlBuildPipeline(ctx);
  owlBuildSBT(ctx);
  std::vector<char> out;
  auto append = [&](OWLBuffer ob, OWLBuffer tb) {
    std::vector<int> prim(4 * n);
    CUDA_CHECK(cudaMemcpy(prim.data(), owlBufferGetPointer(ob, 0), prim.size() * 4, cudaMemcpyDeviceToHost));
    const float *t = (const float *)owlBufferGetPointer(tb, 0);  // host-pinned: read where it lies
    out.insert(out.end(), (const char *)prim.data(), (const char *)(prim.data() + prim.size()));
    out.insert(out.end(), (const char *)t, (const char *)(t + n));
  };
  // pass A: two OWLParams, launched asynchronously back to back on the same raygen, then both waited for --
  // each launch must see its own parameters (the code object has ONE `optixLaunchParams`)
  owlAsyncLaunch2D(rg, n, 1, lpA);
  owlAsyncLaunch2D(rg, n, 1, lpB);
  owlLaunchSync(lpA);
  owlLaunchSync(lpB);
  append(outA, tA);
  append(outB, tB);
  // pass B: the any-hit program rejects odd primitives (the geometry's variable changes: SBT rebuilt)
  owlGeomSet1i(geom, "reject_odd", 1);
  owlBuildSBT(ctx);
  owlLaunch2D(rg, n, 1, lpA);
  append(outA, tA);
  write_file(argv[3], out.data(), out.size());
  std::printf("device_count=%d\n", device_count);
  owlContextDestroy(ctx);
  return 0;
}

static int checks = 0, failures = 0;
static void expect_throw(const char *what, const std::function<void()> &f, const char *needle = nullptr) {
  checks++;
  try {
    f();
    failures++;
    std::printf("FAIL %s: no exception\n", what);
  } catch (const std::runtime_error &e) {
    if (needle && !std::strstr(e.what(), needle)) {
      failures++;
      std::printf("FAIL %s: message '%s' lacks '%s'\n", what, e.what(), needle);
    } else {
      std::printf("PASS %s (%s)\n", what, e.what());
    }
  }
}

static int run_errors(int argc, char **argv) {
  if (argc < 3) return 2;
  std::vector<char> code = read_file(argv[2]);
  if (argc >= 4) {
    // a module built against another revision of owl/device_runtime.h (its layout word differs from the library's): refused
    // when it is loaded, not launched with its arguments at the wrong offsets
    std::vector<char> stale = read_file(argv[3]);
    OWLContext c2 = owlContextCreate(nullptr, 1);
    OWLModule m2 = owlModuleCreate(c2, stale.data());
    OWLVarDecl gv[] = {{"centers", OWL_BUFPTR, 0}, {"radius", OWL_FLOAT, 8}, {nullptr, OWL_INVALID_TYPE, 0}};
    OWLGeomType t2 = owlGeomTypeCreate(c2, OWL_GEOMETRY_USER, 16, gv, -1);
    owlGeomTypeSetBoundsProg(t2, m2, "Balls");
    expect_throw("device code of another header revision", [&] { owlBuildPrograms(c2); }, "another revision");
    owlContextDestroy(c2);
  }
  OWLContext ctx = owlContextCreate(nullptr, 1);
  OWLModule mod = owlModuleCreate(ctx, code.data());
  OWLVarDecl geomVars[] = {{"centers", OWL_BUFPTR, 0}, {"radius", OWL_FLOAT, 8}, {nullptr, OWL_INVALID_TYPE, 0}};
  OWLGeomType type = owlGeomTypeCreate(ctx, OWL_GEOMETRY_USER, 16, geomVars, -1);
  owlGeomTypeSetIntersectProg(type, 0, mod, "Balls");
  owlGeomTypeSetBoundsProg(type, mod, "Balls");
  OWLGeom geom = owlGeomCreate(ctx, type);
  Point3 one[4] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  OWLBuffer buf = owlDeviceBufferCreate(ctx, OWL_USER_TYPE(Point3), 4, one);
  owlGeomSetPrimCount(geom, 4);
  owlGeomSetBuffer(geom, "centers", buf);
  owlGeomSet1f(geom, "radius", 0.5f);
  OWLGroup blas = owlUserGeomGroupCreate(ctx, 1, &geom);
  // impl.cpp:222-224 unknown variable; Variable.cpp:27-36 type mismatch
  expect_throw("unknown variable name", [&] { owlGeomSet1f(geom, "nope", 1.f); }, "could not find variable");
  expect_throw("type mismatch", [&] { owlGeomSet1i(geom, "radius", 1); }, "mismatch");
  expect_throw("buffer set on a float variable", [&] { owlGeomSetBuffer(geom, "radius", buf); });
  // APIHandle.h:57-72 wrong handle kind
  expect_throw("wrong handle kind", [&] { owlGeomSetPrimCount((OWLGeom)buf, 3); }, "expected");
  // UserGeom.cu:213-216 build before owlBuildPrograms; UserGeomGroup.cpp:75-76 refit before build
  expect_throw("accel build before owlBuildPrograms", [&] { owlGroupBuildAccel(blas); }, "owlBuildPrograms");
  owlBuildPrograms(ctx);
  expect_throw("refit before build", [&] { owlGroupRefitAccel(blas); }, "refit before build");
  owlGroupBuildAccel(blas);
  owlGroupRefitAccel(blas);
  checks++;
  std::printf("PASS build then refit\n");
  // Variable.cpp:336-341 only instance groups can be traced
  OWLVarDecl rgVars[] = {{"world", OWL_GROUP, 0}, {"n_queries", OWL_INT, 8}, {nullptr, OWL_INVALID_TYPE, 0}};
  OWLRayGen rg = owlRayGenCreate(ctx, mod, "queries", 16, rgVars, -1);
  expect_throw("geometry group on OWL_GROUP variable", [&] { owlRayGenSetGroup(rg, "world", blas); }, "instance group");
  expect_throw("missing program", [&] {
    OWLRayGen bad = owlRayGenCreate(ctx, mod, "doesNotExist", 16, rgVars, -1);
    (void)bad;
    owlBuildPrograms(ctx);
  }, "not found");
  expect_throw("unsupported subsystem says so", [&] { owlTrianglesGeomGroupCreate(ctx, 0, nullptr); }, "not supported");
  // variable handles
  OWLVariable v = owlGeomGetVariable(geom, "radius");
  owlVariableSet1f(v, 0.25f);
  owlVariableRelease(v);
  checks++;
  std::printf("PASS variable handle set/release\n");
  owlContextDestroy(ctx);
  std::printf("checks=%d failures=%d\n", checks, failures);
  return failures ? 1 : 0;
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  try {
    if (!std::strcmp(argv[1], "knn")) return run_knn(argc, argv);
    if (!std::strcmp(argv[1], "count")) return run_count(argc, argv);
    if (!std::strcmp(argv[1], "errors")) return run_errors(argc, argv);
    if (!std::strcmp(argv[1], "api")) return run_api(argc, argv);
  } catch (const std::exception &e) {
    std::fprintf(stderr, "uncaught: %s\n", e.what());
    return 3;
  }
  return 2;
}
