// hostCode.cpp -- sample02-rtdbscan: RT-DBSCAN as an OWL application, in the shape of
// samples/s01-trueknn/hostCode.cpp (CSV in, one geometry of n custom primitives, launches driven
// from the host).  Only owl* calls and the CUDA-named runtime calls the OWL headers bring.
//
//   sample02-rtdbscan <points.csv> <nPoints> <dim 2|3> <eps> <minPts> <out.bin>
//
// out.bin: n int32 labels (clusters numbered by ascending smallest core index, noise -1), then n
// bytes of core flags.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "GeomTypes.h"

extern "C" char ptxCode[];

static double seconds_since(const std::chrono::steady_clock::time_point &t0) {
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

int main(int argc, char **argv) {
  if (argc < 7) {
    std::cerr << "usage: " << argv[0] << " <points.csv> <nPoints> <dim> <eps> <minPts> <out.bin>\n";
    return 2;
  }
  const size_t n = (size_t)std::atoll(argv[2]);
  const int dim = std::atoi(argv[3]);
  const float eps = (float)std::atof(argv[4]);
  const int minPts = std::atoi(argv[5]);
  // ---- the point set: comma-separated rows, as samples/s01-trueknn reads them
  std::vector<float> values;
  {
    std::ifstream file(argv[1]);
    if (!file) {
      std::cerr << "cannot open " << argv[1] << "\n";
      return 2;
    }
    std::string line;
    while (values.size() < n * (size_t)dim && std::getline(file, line)) {
      std::stringstream ss(line);
      float f;
      while (ss >> f) {
        values.push_back(f);
        if (ss.peek() == ',') ss.ignore();
      }
    }
  }
  if (values.size() < n * (size_t)dim) {
    std::cerr << "file holds fewer than " << n << " points\n";
    return 2;
  }
  std::vector<Point> points(n);
  for (size_t i = 0; i < n; i++)
    points[i].center = vec3f(values[i * dim], values[i * dim + 1], dim == 3 ? values[i * dim + 2] : 0.f);  // 2-D: z = 0

  // ---- OWL set-up
  OWLContext context = owlContextCreate(nullptr, 1);
  OWLModule module = owlModuleCreate(context, ptxCode);
  OWLVarDecl geomVars[] = {{"prims", OWL_BUFPTR, OWL_OFFSETOF(PointsGeom, prims)},
                           {"eps", OWL_FLOAT, OWL_OFFSETOF(PointsGeom, eps)},
                           {/* sentinel */}};
  OWLGeomType geomType = owlGeomTypeCreate(context, OWL_GEOMETRY_USER, sizeof(PointsGeom), geomVars, -1);
  owlGeomTypeSetIntersectProg(geomType, 0, module, "Points");
  owlGeomTypeSetBoundsProg(geomType, module, "Points");
  owlBuildPrograms(context);

  OWLBuffer pointsBuffer = owlDeviceBufferCreate(context, OWL_USER_TYPE(Point), n, points.data());
  OWLBuffer countBuffer = owlDeviceBufferCreate(context, OWL_INT, n, nullptr);
  OWLBuffer coreBuffer = owlManagedMemoryBufferCreate(context, OWL_UCHAR, n, nullptr);
  OWLBuffer parentBuffer = owlDeviceBufferCreate(context, OWL_INT, n, nullptr);
  OWLBuffer labelBuffer = owlManagedMemoryBufferCreate(context, OWL_INT, n, nullptr);
  {
    std::vector<int> zeros(n, 0);
    owlBufferUpload(countBuffer, zeros.data(), 0, n * sizeof(int));
  }

  OWLGeom geom = owlGeomCreate(context, geomType);
  owlGeomSetPrimCount(geom, n);
  owlGeomSetBuffer(geom, "prims", pointsBuffer);
  owlGeomSet1f(geom, "eps", eps);

  OWLVarDecl globalsVars[] = {{"points", OWL_BUFPTR, OWL_OFFSETOF(Globals, points)},
                              {"eps", OWL_FLOAT, OWL_OFFSETOF(Globals, eps)},
                              {"minPts", OWL_INT, OWL_OFFSETOF(Globals, minPts)},
                              {"phase", OWL_INT, OWL_OFFSETOF(Globals, phase)},
                              {"count", OWL_BUFPTR, OWL_OFFSETOF(Globals, count)},
                              {"core", OWL_BUFPTR, OWL_OFFSETOF(Globals, core)},
                              {"parent", OWL_BUFPTR, OWL_OFFSETOF(Globals, parent)},
                              {"label", OWL_BUFPTR, OWL_OFFSETOF(Globals, label)},
                              {/* sentinel */}};
  OWLParams lp = owlParamsCreate(context, sizeof(Globals), globalsVars, -1);
  owlParamsSetBuffer(lp, "points", pointsBuffer);
  owlParamsSet1f(lp, "eps", eps);
  owlParamsSet1i(lp, "minPts", minPts);
  owlParamsSetBuffer(lp, "count", countBuffer);
  owlParamsSetBuffer(lp, "core", coreBuffer);
  owlParamsSetBuffer(lp, "parent", parentBuffer);
  owlParamsSetBuffer(lp, "label", labelBuffer);

  auto t0 = std::chrono::steady_clock::now();
  OWLGroup pointsGroup = owlUserGeomGroupCreate(context, 1, &geom);
  owlGroupBuildAccel(pointsGroup);
  OWLGroup world = owlInstanceGroupCreate(context, 1, &pointsGroup);
  owlGroupBuildAccel(world);
  std::cout << "Build time: " << seconds_since(t0) << std::endl;

  OWLVarDecl rayGenVars[] = {{"world", OWL_GROUP, OWL_OFFSETOF(RayGenData, world)},
                             {"n", OWL_INT, OWL_OFFSETOF(RayGenData, n)},
                             {/* sentinel */}};
  OWLRayGen rayGen = owlRayGenCreate(context, module, "rayGen", sizeof(RayGenData), rayGenVars, -1);
  owlRayGenSetGroup(rayGen, "world", world);
  owlRayGenSet1i(rayGen, "n", (int)n);
  owlBuildPrograms(context);
  owlBuildPipeline(context);
  owlBuildSBT(context);

  // ---- the clustering: five launches, the phase in the launch parameters
  t0 = std::chrono::steady_clock::now();
  const int phases[] = {DB_COUNT, DB_MARK, DB_UNION, DB_FLATTEN, DB_BORDER};
  const char *names[] = {"neighbour counts", "core flags", "unions", "roots", "border points"};
  for (int i = 0; i < 5; i++) {
    auto tp = std::chrono::steady_clock::now();
    owlParamsSet1i(lp, "phase", phases[i]);
    owlLaunch2D(rayGen, (int)n, 1, lp);
    std::cout << "Phase " << i + 1 << " (" << names[i] << ") time: " << seconds_since(tp) << std::endl;
  }
  // labels hold the root (= smallest core index) of each point's cluster, or -1: number the clusters
  // by ascending root (the spec's order)
  const int *roots = (const int *)owlBufferGetPointer(labelBuffer, 0);
  const unsigned char *core = (const unsigned char *)owlBufferGetPointer(coreBuffer, 0);
  std::vector<int> distinct;
  for (size_t i = 0; i < n; i++)
    if (core[i] && roots[i] == (int)i) distinct.push_back((int)i);  // a root is its own smallest core index
  std::vector<int> labels(n);
  size_t noise = 0;
  for (size_t i = 0; i < n; i++) {
    labels[i] = roots[i] < 0 ? -1 : (int)(std::lower_bound(distinct.begin(), distinct.end(), roots[i]) - distinct.begin());
    noise += labels[i] < 0;
  }
  std::cout << "RT-DBSCAN time: " << seconds_since(t0) << std::endl;
  std::cout << "clusters=" << distinct.size() << " noise=" << noise << std::endl;

  std::ofstream out(argv[6], std::ios::binary);
  out.write((const char *)labels.data(), (std::streamsize)(n * sizeof(int)));
  out.write((const char *)core, (std::streamsize)n);
  owlContextDestroy(context);
  return 0;
}
