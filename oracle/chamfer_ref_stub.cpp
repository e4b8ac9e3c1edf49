// ORACLE (test infrastructure only).  The reference's chamfer_distance.cpp declares the two CUDA launchers of chamfer_distance.cu;
// the CPU oracle build links these stand-ins instead (the CPU entry points chamfer_distance_forward / _backward never call them).
#include <cstdio>
#include <cstdlib>

int ChamferDistanceKernelLauncher(const int, const int, const float*, const int, const float*, float*, int*, float*, int*) {
  std::fprintf(stderr, "oracle/_ref: the CUDA path of the reference chamfer distance is not part of the CPU oracle\n");
  std::abort();
}
int ChamferDistanceGradKernelLauncher(const int, const int, const float*, const int, const float*, const float*, const int*, const float*,
                                      const int*, float*, float*) {
  std::fprintf(stderr, "oracle/_ref: the CUDA path of the reference chamfer distance is not part of the CPU oracle\n");
  std::abort();
}
