/*
 * cudecomp_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-process CPU restatement of the cuDecomp hot path (pencil index maps, the
 * transpose pack / all-to-all / unpack plan, and the halo update), executed for ALL ranks of a
 * simulated process grid inside one address space (the all-to-all is a memcpy between the
 * per-rank buffers).  It exists to check the HIP product library; nothing in the product may
 * include, link or call it.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it.
 *
 * Every function cites the reference file:line (under /root/reference) whose behaviour it
 * restates.  Parity is pinned: the index maps are checked against the reference's golden
 * vectors (tests/ctest/api_tests.cc:92-153, 1386-1408) and the data movement against the
 * reference's analytic test oracle (tests/ctest/transpose_tests.cc:323-354,
 * tests/ctest/halo_tests.cc:214-253), both restated in this file as well.
 */
#ifndef CUDECOMP_ORACLE_H
#define CUDECOMP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_OK = 0, ORC_INVALID_USAGE = 1, ORC_NOT_SUPPORTED = 2 }; /* cudecompResult_t values, cudecomp.h:102-113 */
enum { ORC_RANK_ORDER_ROW_MAJOR = 1, ORC_RANK_ORDER_COL_MAJOR = 2 }; /* cudecomp.h:91-96 */

#define ORC_MAX_COMM 64

typedef struct {
  int32_t gdims[3];
  int32_t gdims_dist[3]; /* resolved: == gdims when not set by the user */
  int32_t pdims[2];
  int32_t rank_order;      /* resolved (never DEFAULT) */
  int32_t mem_order[3][3]; /* resolved transpose_mem_order[axis][i] */
} orc_grid_t;

/* Same fields as cudecompPencilInfo_t (cudecomp.h:224-238) minus the versioning header. */
typedef struct {
  int32_t shape[3];
  int32_t lo[3];
  int32_t hi[3];
  int32_t order[3];
  int32_t halo_extents[3];
  int32_t padding[3];
  int64_t size;
} orc_pinfo_t;

/* Resolve a grid description the way cudecompGridDescCreate does (src/cudecomp.cc:1120-1150).
 * gdims_dist may be NULL / contain zeros (-> gdims).  mem_order may be NULL or start with a
 * negative entry (-> derived from axis_contiguous).  rank_order 0 resolves to row-major. */
int orc_grid_init(orc_grid_t* g, const int32_t gdims[3], const int32_t gdims_dist[3], const int32_t pdims[2],
                  int32_t rank_order, const int32_t axis_contiguous[3], const int32_t mem_order[9]);

int orc_nranks(const orc_grid_t* g);
void orc_pidx(const orc_grid_t* g, int rank, int32_t pidx[2]);
/* comm_axis: 0 = column communicator (pdims[0] ranks), 1 = row communicator (pdims[1] ranks) */
int orc_global_rank(const orc_grid_t* g, int rank, int comm_axis, int comm_rank);

int orc_pencil_info(const orc_grid_t* g, int rank, int axis, const int32_t halo_extents[3], const int32_t padding[3],
                    orc_pinfo_t* out);
int orc_shifted_rank(const orc_grid_t* g, int rank, int axis, int dim, int displacement, int periodic,
                     int32_t* shifted_rank);
int64_t orc_align_count(int64_t count, int nbytes);
int64_t orc_transpose_workspace_size(const orc_grid_t* g);
int orc_halo_workspace_size(const orc_grid_t* g, int rank, int axis, const int32_t halo_extents[3], int64_t* size);
void orc_get_splits(int64_t n, int nchunks, int pad, int64_t* splits);
void orc_peer_ranks(int nranks, int npergroup, int rank, int iter, int* src_rank, int* dst_rank);

/* Transpose for every rank of the grid.  ax/dir as in transpose.h:907-953: XToY (0,+1),
 * YToZ (1,+1), ZToY (2,-1), YToX (1,-1).  in/out/work are arrays of nranks pointers (element
 * size es bytes: 4, 8 or 16).  in[r] == out[r] selects the in-place plan.  pipelined selects the
 * per-peer plan of the *_PL backends (same results, different staging). */
int orc_transpose(const orc_grid_t* g, int ax, int dir, int es, void* const* in, void* const* out, void* const* work,
                  const int32_t in_halo[3], const int32_t out_halo[3], const int32_t in_pad[3],
                  const int32_t out_pad[3], int pipelined);

/* Halo update along `dim` of the axis-`ax` pencils of every rank (halo.h:41-315).
 * staged != 0 forces the packed path even where faces are contiguous (what NVSHMEM-class
 * transports do, halo.h:150-162). */
int orc_update_halos(const orc_grid_t* g, int ax, int es, void* const* data, void* const* work,
                     const int32_t halo_extents[3], const int32_t periods[3], int dim, const int32_t padding[3],
                     int staged);

/* The reference's analytic test oracle.  Values are float64-representable global linear indices
 * gx + X*(gy + Y*gz) written as the dtype (kind: 0 f32, 1 f64, 2 complex f32 (g,-g), 3 complex
 * f64 (g,-g)); -1 (complex: (-1,-1) for transposes, (-1,0) for halos) elsewhere. */
void orc_fill_pencil(const orc_pinfo_t* p, const int32_t gdims[3], int kind, int halo_style, void* data);
void orc_fill_halo_reference(const orc_pinfo_t* p, const int32_t gdims[3], const int32_t periods[3], int kind,
                             void* data);
/* 0 if equal (interior_only: ignore halo/padding cells), else 1 + index of the first mismatch. */
int64_t orc_compare_pencil(const orc_pinfo_t* p, int es, const void* expected, const void* actual, int interior_only);

/* exchange callback of orc_transpose_rank: chunk d of `send` (send_cnt[d] elements at send_off[d]) goes to member d
 * of the row (comm_axis 1) / column (0) communicator, chunk s of `recv` arrives from member s; counts in elements */
typedef void (*orc_exchange_fn)(void* user, const char* send, const int64_t* send_cnt, const int64_t* send_off, char* recv,
                                const int64_t* recv_cnt, const int64_t* recv_off, int nmembers, int comm_axis,
                                int comm_rank, int es);
int orc_transpose_rank(const orc_grid_t* g, int rank, int ax, int dir, int es, void* in, void* out, void* work,
                       const int32_t in_halo[3], const int32_t out_halo[3], const int32_t in_pad[3],
                       const int32_t out_pad[3], int pipelined, orc_exchange_fn exchange, void* user);

#ifdef __cplusplus
}
#endif
#endif
