/*
 * basic_usage.c -- a plain C solver skeleton written against cudecomp.h, the way a cuDecomp user's code
 * looks: describe the grid, ask for pencils and workspace, run the X->Y->Z->Y->X transposes and a halo
 * update.  It is the link-compatibility check of the drop-in boundary: nothing here is specific to this
 * implementation except that device memory comes from HIP.
 *
 *   hipcc -x c -Iinclude examples/c/basic_usage.c -Lcudecomp_amd/lib -lcudecomp -o basic_usage
 *   ./basic_usage                                  (one rank)
 *   RANK=r WORLD_SIZE=n ./basic_usage  (n processes; or mpirun -np n with -DUSE_MPI and libcudecomp_mpi)
 *
 * Every cell carries its global linear index; after each transpose the local pencil is checked against the
 * closed form, so the program verifies what it demonstrates.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef USE_MPI
#include <mpi.h>
#endif
#include <hip/hip_runtime_api.h>

#include "cudecomp.h"

#define CHECK_CUDECOMP(call)                                                        \
  do {                                                                              \
    cudecompResult_t r_ = (call);                                                   \
    if (r_ != CUDECOMP_RESULT_SUCCESS) {                                            \
      fprintf(stderr, "%s:%d cuDecomp error %d\n", __FILE__, __LINE__, (int)r_);    \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)
#define CHECK_HIP(call)                                                             \
  do {                                                                              \
    hipError_t e_ = (call);                                                         \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s:%d HIP error %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

static const int32_t gdims[3] = {48, 40, 56};

/* fill (check = 0) or verify (check = 1) the interior of a pencil with global linear indices */
static long pencil_values(const cudecompPencilInfo_t* p, double* data, int check) {
  long bad = 0;
  for (int64_t i = 0; i < p->size; ++i) {
    int64_t l[3] = {i % p->shape[0], i / p->shape[0] % p->shape[1], i / ((int64_t)p->shape[0] * p->shape[1])};
    int64_t g[3];
    int interior = 1;
    for (int k = 0; k < 3; ++k) {
      int ax = p->order[k];
      if (l[k] < p->halo_extents[ax] || l[k] >= p->shape[k] - p->halo_extents[ax] - p->padding[ax]) interior = 0;
      g[ax] = l[k] + p->lo[k] - p->halo_extents[ax];
    }
    if (!interior) {
      if (!check) data[i] = -1.0;
      continue;
    }
    double v = (double)(g[0] + gdims[0] * (g[1] + (int64_t)gdims[1] * g[2]));
    if (check) bad += (data[i] != v);
    else data[i] = v;
  }
  return bad;
}

int main(int argc, char** argv) {
  int rank = 0, nranks = 1;
#ifdef USE_MPI
  MPI_Init(&argc, &argv);
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &nranks);
#else
  (void)argc;
  (void)argv;
  if (getenv("RANK")) rank = atoi(getenv("RANK"));
  if (getenv("WORLD_SIZE")) nranks = atoi(getenv("WORLD_SIZE"));
#endif
  int ndev = 0;
  CHECK_HIP(hipGetDeviceCount(&ndev));
  CHECK_HIP(hipSetDevice(rank % ndev));

  cudecompHandle_t handle;
  CHECK_CUDECOMP(cudecompInit(&handle, MPI_COMM_WORLD));

  cudecompGridDescConfig_t config;
  CHECK_CUDECOMP(cudecompGridDescConfigSetDefaults(&config));
  for (int i = 0; i < 3; ++i) config.gdims[i] = gdims[i];
  config.pdims[0] = (nranks % 2 == 0 && nranks > 2) ? 2 : 1;
  config.pdims[1] = nranks / config.pdims[0];
  for (int i = 0; i < 3; ++i) config.transpose_axis_contiguous[i] = true;
  if (getenv("EXAMPLE_TRANSPOSE_BACKEND")) config.transpose_comm_backend = atoi(getenv("EXAMPLE_TRANSPOSE_BACKEND"));
  if (getenv("EXAMPLE_HALO_BACKEND")) config.halo_comm_backend = atoi(getenv("EXAMPLE_HALO_BACKEND"));

  cudecompGridDesc_t grid_desc;
  CHECK_CUDECOMP(cudecompGridDescCreate(handle, &grid_desc, &config, NULL));
  if (rank == 0)
    printf("grid %d x %d x %d on a %d x %d process grid, transpose backend %s, halo backend %s\n", gdims[0], gdims[1],
           gdims[2], config.pdims[0], config.pdims[1], cudecompTransposeCommBackendToString(config.transpose_comm_backend),
           cudecompHaloCommBackendToString(config.halo_comm_backend));

  const int32_t halo[3] = {1, 1, 1};
  const bool periods[3] = {true, true, true};
  cudecompPencilInfo_t px, py, pz, pxh;
  CHECK_CUDECOMP(cudecompGetPencilInfo(handle, grid_desc, &px, 0, NULL, NULL));
  CHECK_CUDECOMP(cudecompGetPencilInfo(handle, grid_desc, &py, 1, NULL, NULL));
  CHECK_CUDECOMP(cudecompGetPencilInfo(handle, grid_desc, &pz, 2, NULL, NULL));
  CHECK_CUDECOMP(cudecompGetPencilInfo(handle, grid_desc, &pxh, 0, halo, NULL));

  int64_t nel = px.size, ws_t = 0, ws_h = 0, dsize = 0;
  if (py.size > nel) nel = py.size;
  if (pz.size > nel) nel = pz.size;
  if (pxh.size > nel) nel = pxh.size;
  CHECK_CUDECOMP(cudecompGetTransposeWorkspaceSize(handle, grid_desc, &ws_t));
  CHECK_CUDECOMP(cudecompGetHaloWorkspaceSize(handle, grid_desc, 0, halo, &ws_h));
  CHECK_CUDECOMP(cudecompGetDataTypeSize(CUDECOMP_DOUBLE, &dsize));
  int64_t ws = ws_t > ws_h ? ws_t : ws_h;

  double* host = (double*)malloc((size_t)nel * sizeof(double));
  double *data_d, *work_d;
  CHECK_HIP(hipMalloc((void**)&data_d, (size_t)nel * dsize));
  CHECK_CUDECOMP(cudecompMalloc(handle, grid_desc, (void**)&work_d, (size_t)ws * dsize));

  long bad = 0;
  pencil_values(&px, host, 0);
  CHECK_HIP(hipMemcpy(data_d, host, (size_t)px.size * dsize, hipMemcpyHostToDevice));

  /* in-place transposes on the default stream */
  CHECK_CUDECOMP(cudecompTransposeXToY(handle, grid_desc, data_d, data_d, work_d, CUDECOMP_DOUBLE, NULL, NULL, NULL, NULL, 0));
  CHECK_HIP(hipMemcpy(host, data_d, (size_t)py.size * dsize, hipMemcpyDeviceToHost));
  bad += pencil_values(&py, host, 1);
  CHECK_CUDECOMP(cudecompTransposeYToZ(handle, grid_desc, data_d, data_d, work_d, CUDECOMP_DOUBLE, NULL, NULL, NULL, NULL, 0));
  CHECK_HIP(hipMemcpy(host, data_d, (size_t)pz.size * dsize, hipMemcpyDeviceToHost));
  bad += pencil_values(&pz, host, 1);
  CHECK_CUDECOMP(cudecompTransposeZToY(handle, grid_desc, data_d, data_d, work_d, CUDECOMP_DOUBLE, NULL, NULL, NULL, NULL, 0));
  CHECK_HIP(hipMemcpy(host, data_d, (size_t)py.size * dsize, hipMemcpyDeviceToHost));
  bad += pencil_values(&py, host, 1);
  /* back to X pencils, this time into a halo-carrying layout, then fill the halos */
  CHECK_CUDECOMP(cudecompTransposeYToX(handle, grid_desc, data_d, data_d, work_d, CUDECOMP_DOUBLE, NULL, halo, NULL, NULL, 0));
  for (int dim = 0; dim < 3; ++dim)
    CHECK_CUDECOMP(cudecompUpdateHalosX(handle, grid_desc, data_d, work_d, CUDECOMP_DOUBLE, halo, periods, dim, NULL, 0));
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpy(host, data_d, (size_t)pxh.size * dsize, hipMemcpyDeviceToHost));
  bad += pencil_values(&pxh, host, 1);
  /* a periodic halo cell equals the wrapped interior value: check the cell left of the first interior x */
  {
    int64_t off = (int64_t)pxh.shape[0] * (halo[1] + (int64_t)pxh.shape[1] * halo[2]); /* x fastest: order {0,1,2} */
    double expect = (double)((gdims[0] - 1) + gdims[0] * (pxh.lo[1] + (int64_t)gdims[1] * pxh.lo[2]));
    bad += (host[off] != expect);
  }

  printf("rank %d: %s (%ld mismatches)\n", rank, bad == 0 ? "PASSED" : "FAILED", bad);

  CHECK_CUDECOMP(cudecompFree(handle, grid_desc, work_d));
  CHECK_HIP(hipFree(data_d));
  free(host);
  CHECK_CUDECOMP(cudecompGridDescDestroy(handle, grid_desc));
  CHECK_CUDECOMP(cudecompFinalize(handle));
#ifdef USE_MPI
  MPI_Finalize();
#endif
  return bad != 0;
}
