/* cpu_mpi_cycle.c -- the CPU baseline of bench.py: the X->Y->Z->Y->X transpose cycle on HOST memory with one MPI rank
 * per core -- pack -> MPI_Alltoallv inside the row / column communicator -> unpack, the phases of the reference's
 * transpose path (include/internal/transpose.h:196-905; exchange semantics comm_routines.h:363-413) executed by the
 * oracle's own pack / unpack code (orc_transpose_rank).  The reference itself has no CPU path (its benchmark is GPU
 * only, benchmark/benchmark.cu:139, 435-447); this is the "host-MPI CPU path" the measurement is quoted next to.
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY -- nothing in the product links or calls it.
 *
 *   mpirun -np R ./cpu_mpi_cycle N prow pcol contiguous(0|1) warmup trials [kind: 0 fp32 | 1 fp64 (default) | 2 c64 | 3 c128]
 * prints one JSON line on rank 0: cycle time (max over ranks, min / max / avg / std over the trials, MPI_Wtime after a
 * barrier as benchmark.cu:503-505, 587-590 times the GPU path) and effective GB/s = 4 * N^3 * element bytes / t.
 * The line also carries the MPI library's version string, how many distinct cores the ranks sit on and the bytes each rank
 * holds.  BASELINE config 1 is `mpirun -np 2 ./cpu_mpi_cycle 256 2 1 0 3 5 0` (256^3 fp32 slab, 2 ranks). */
#define _GNU_SOURCE
#include <math.h>
#include <mpi.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cudecomp_oracle.h"

typedef struct {
  MPI_Comm row, col;
} comms_t;

static void exchange(void* user, const char* send, const int64_t* send_cnt, const int64_t* send_off, char* recv,
                     const int64_t* recv_cnt, const int64_t* recv_off, int P, int comm_axis, int comm_rank, int es) {
  (void)comm_rank;
  comms_t* c = (comms_t*)user;
  int sc[ORC_MAX_COMM], so[ORC_MAX_COMM], rc[ORC_MAX_COMM], ro[ORC_MAX_COMM];
  /* counts in words of min(es, 8) bytes keep 32-bit MPI counts in range for the sample sizes */
  const int w = es < 8 ? es : 8;
  for (int i = 0; i < P; ++i) {
    sc[i] = (int)(send_cnt[i] * es / w);
    so[i] = (int)(send_off[i] * es / w);
    rc[i] = (int)(recv_cnt[i] * es / w);
    ro[i] = (int)(recv_off[i] * es / w);
  }
  const MPI_Datatype t = (w == 4) ? MPI_FLOAT : MPI_DOUBLE;
  MPI_Alltoallv(send, sc, so, t, recv, rc, ro, t, comm_axis == 0 ? c->col : c->row);
}

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  int rank, nranks;
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &nranks);
  if (argc < 7) {
    if (rank == 0) fprintf(stderr, "usage: cpu_mpi_cycle N prow pcol contiguous warmup trials\n");
    MPI_Finalize();
    return 2;
  }
  const int n = atoi(argv[1]), pr = atoi(argv[2]), pc = atoi(argv[3]), contiguous = atoi(argv[4]);
  const int warmup = atoi(argv[5]), trials = atoi(argv[6]);
  const int kind = argc > 7 ? atoi(argv[7]) : 1;
  static const int kind_bytes[4] = {4, 8, 8, 16};
  if (kind < 0 || kind > 3) {
    if (rank == 0) fprintf(stderr, "kind must be 0..3\n");
    MPI_Finalize();
    return 2;
  }
  if (pr * pc != nranks) {
    if (rank == 0) fprintf(stderr, "process grid %d x %d does not match %d ranks\n", pr, pc, nranks);
    MPI_Finalize();
    return 2;
  }
  const int32_t gdims[3] = {n, n, n}, pdims[2] = {pr, pc};
  const int32_t ac[3] = {contiguous, contiguous, contiguous};
  orc_grid_t g;
  if (orc_grid_init(&g, gdims, NULL, pdims, 1 /* row-major */, ac, NULL) != ORC_OK) return 3;

  /* communicators as the library builds them: column = same pidx[1] (spans pdims[0]), row = same pidx[0] */
  int32_t pidx[2];
  orc_pidx(&g, rank, pidx);
  comms_t comms;
  MPI_Comm_split(MPI_COMM_WORLD, pidx[1], pidx[0], &comms.col);
  MPI_Comm_split(MPI_COMM_WORLD, pidx[0], pidx[1], &comms.row);

  orc_pinfo_t p[3];
  int64_t nel = 0;
  for (int ax = 0; ax < 3; ++ax) {
    orc_pencil_info(&g, rank, ax, NULL, NULL, &p[ax]);
    if (p[ax].size > nel) nel = p[ax].size;
  }
  const int es = kind_bytes[kind];
  const int64_t ws = orc_transpose_workspace_size(&g);
  char* a = (char*)malloc((size_t)nel * es);
  char* b = (char*)malloc((size_t)nel * es);
  char* w = (char*)malloc((size_t)(ws > 0 ? ws : 1) * es);
  if (!a || !b || !w) return 4;
  orc_fill_pencil(&p[0], gdims, kind, 0, a);
  memset(b, 0, (size_t)nel * es);
  memset(w, 0, (size_t)(ws > 0 ? ws : 1) * es);
  char* ref = (char*)malloc((size_t)p[0].size * es);
  if (!ref) return 4;
  memcpy(ref, a, (size_t)p[0].size * es);

  const int ax_of[4] = {0, 1, 2, 1}, dir_of[4] = {+1, +1, -1, -1};
  double total = 0, tmin = 1e30, tmax = 0, sq = 0;
  for (int t = 0; t < warmup + trials; ++t) {
    MPI_Barrier(MPI_COMM_WORLD);
    const double t0 = MPI_Wtime();
    char *cur = a, *nxt = b;
    for (int op = 0; op < 4; ++op) {
      if (orc_transpose_rank(&g, rank, ax_of[op], dir_of[op], es, cur, nxt, w, NULL, NULL, NULL, NULL, 0, exchange,
                             &comms) != ORC_OK) {
        fprintf(stderr, "rank %d: transpose %d failed\n", rank, op);
        MPI_Abort(MPI_COMM_WORLD, 5);
      }
      char* tmp = cur;
      cur = nxt;
      nxt = tmp;
    }
    double dt = MPI_Wtime() - t0, dmax = 0;
    MPI_Allreduce(&dt, &dmax, 1, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD);
    if (t >= warmup) {
      total += dmax;
      sq += dmax * dmax;
      if (dmax < tmin) tmin = dmax;
      if (dmax > tmax) tmax = dmax;
    }
  }
  /* four hops, out of place: the X pencil must be back in `a`, bit for bit */
  int ok = memcmp(ref, a, (size_t)p[0].size * es) == 0, all_ok = 0;
  free(ref);
  MPI_Allreduce(&ok, &all_ok, 1, MPI_INT, MPI_MIN, MPI_COMM_WORLD);
  /* provenance (CPU model, cores, MPI version next to the number): the MPI library as it names itself, and
   * the core every rank sits on now (with `mpirun -bind-to core` these are distinct) */
  char version[MPI_MAX_LIBRARY_VERSION_STRING] = {0};
  int vlen = 0;
  MPI_Get_library_version(version, &vlen);
  for (int i = 0; version[i]; ++i) {
    if (version[i] == '\n' || version[i] == '\r') {
      version[i] = 0; /* first line only */
      break;
    }
    if (version[i] == '"' || version[i] == '\\' || (unsigned char)version[i] < 32) version[i] = ' ';
  }
  int my_cpu = sched_getcpu();
  int* cpus = (int*)malloc(sizeof(int) * (size_t)nranks);
  MPI_Gather(&my_cpu, 1, MPI_INT, cpus, 1, MPI_INT, 0, MPI_COMM_WORLD);
  if (rank == 0) {
    int distinct = 0;
    for (int i = 0; i < nranks; ++i) {
      int seen = 0;
      for (int j = 0; j < i; ++j) seen |= cpus[j] == cpus[i];
      distinct += !seen;
    }
    const double sec = total / trials;
    const double var = sq / trials - sec * sec;
    printf("{\"n\": %d, \"ranks\": %d, \"pdims\": [%d, %d], \"contiguous\": %d, \"warmup\": %d, \"trials\": %d, "
           "\"element_bytes\": %d, \"cycle_s\": %.6f, \"cycle_s_min\": %.6f, \"cycle_s_max\": %.6f, \"cycle_s_std\": %.6f, "
           "\"gbps\": %.4f, \"round_trip_ok\": %s, \"mpi_version\": \"%.100s\", \"distinct_cpus\": %d, "
           "\"bytes_per_rank\": %lld}\n",
           n, nranks, pr, pc, contiguous, warmup, trials, es, sec, tmin, tmax, var > 0 ? sqrt(var) : 0.0,
           4.0 * n * (double)n * n * es / sec / 1e9, all_ok ? "true" : "false", version, distinct,
           (long long)((2 * nel + (ws > 0 ? ws : 1)) * es));
  }
  free(cpus);
  free(a);
  free(b);
  free(w);
  MPI_Finalize();
  return all_ok ? 0 : 6;
}
