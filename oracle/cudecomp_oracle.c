/*
 * cudecomp_oracle.c -- TEST INFRASTRUCTURE ONLY (see cudecomp_oracle.h).
 *
 * CPU restatement of the cuDecomp transpose / halo path.  Citations are file:line under
 * /root/reference.  The product (cudecomp_amd/csrc) does not share any code with this file.
 */
#include "cudecomp_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* grid description                                                                            */
/* ------------------------------------------------------------------------------------------ */

int orc_grid_init(orc_grid_t* g, const int32_t gdims[3], const int32_t gdims_dist[3], const int32_t pdims[2],
                  int32_t rank_order, const int32_t axis_contiguous[3], const int32_t mem_order[9]) {
  memset(g, 0, sizeof(*g));
  for (int i = 0; i < 3; ++i) g->gdims[i] = gdims[i];
  g->pdims[0] = pdims[0];
  g->pdims[1] = pdims[1];
  if (pdims[0] <= 0 || pdims[1] <= 0) return ORC_INVALID_USAGE;
  if (pdims[0] > ORC_MAX_COMM || pdims[1] > ORC_MAX_COMM) return ORC_NOT_SUPPORTED;
  /* rank order: DEFAULT resolves to row-major (src/cudecomp.cc:725-729) */
  g->rank_order = (rank_order == ORC_RANK_ORDER_COL_MAJOR) ? ORC_RANK_ORDER_COL_MAJOR : ORC_RANK_ORDER_ROW_MAJOR;

  /* gdims_dist: used only if all three entries are non-zero (src/cudecomp.cc:1135-1150) */
  if (gdims_dist) {
    for (int i = 0; i < 3; ++i)
      if (gdims_dist[i] > gdims[i]) return ORC_INVALID_USAGE;
  }
  if (gdims_dist && gdims_dist[0] != 0 && gdims_dist[1] != 0 && gdims_dist[2] != 0) {
    for (int i = 0; i < 3; ++i) g->gdims_dist[i] = gdims_dist[i];
  } else {
    for (int i = 0; i < 3; ++i) g->gdims_dist[i] = gdims[i];
  }

  /* memory order: explicit transpose_mem_order wins, else (axis+i)%3 for axis-contiguous
   * pencils and i otherwise (src/cudecomp.cc:1120-1133) */
  if (mem_order && mem_order[0] >= 0) {
    for (int a = 0; a < 3; ++a) {
      int seen[3] = {0, 0, 0};
      for (int i = 0; i < 3; ++i) {
        int v = mem_order[a * 3 + i];
        if (v < 0 || v > 2 || seen[v]) return ORC_INVALID_USAGE; /* src/cudecomp.cc:460-480 */
        seen[v] = 1;
        g->mem_order[a][i] = v;
      }
    }
  } else {
    for (int a = 0; a < 3; ++a)
      for (int i = 0; i < 3; ++i)
        g->mem_order[a][i] = (axis_contiguous && axis_contiguous[a]) ? (a + i) % 3 : i;
  }
  return ORC_OK;
}

int orc_nranks(const orc_grid_t* g) { return g->pdims[0] * g->pdims[1]; }

/* include/internal/common.h:318-331 */
void orc_pidx(const orc_grid_t* g, int rank, int32_t pidx[2]) {
  if (g->rank_order == ORC_RANK_ORDER_COL_MAJOR) {
    pidx[0] = rank % g->pdims[0];
    pidx[1] = rank / g->pdims[0];
  } else {
    pidx[0] = rank / g->pdims[1];
    pidx[1] = rank % g->pdims[1];
  }
}

/* include/internal/common.h:334-346 (comm_axis 1 == CUDECOMP_COMM_ROW, 0 == CUDECOMP_COMM_COL) */
int orc_global_rank(const orc_grid_t* g, int rank, int comm_axis, int comm_rank) {
  int32_t pidx[2];
  orc_pidx(g, rank, pidx);
  if (g->rank_order == ORC_RANK_ORDER_COL_MAJOR) {
    return (comm_axis == 1) ? pidx[0] + comm_rank * g->pdims[0] : g->pdims[0] * pidx[1] + comm_rank;
  }
  return (comm_axis == 1) ? g->pdims[1] * pidx[0] + comm_rank : pidx[1] + comm_rank * g->pdims[1];
}

/* ------------------------------------------------------------------------------------------ */
/* index maps                                                                                  */
/* ------------------------------------------------------------------------------------------ */

static int64_t min64(int64_t a, int64_t b) { return a < b ? a : b; }

/* src/cudecomp.cc:1317-1379 */
int orc_pencil_info(const orc_grid_t* g, int rank, int axis, const int32_t halo_extents[3], const int32_t padding[3],
                    orc_pinfo_t* p) {
  if (axis < 0 || axis > 2) return ORC_INVALID_USAGE;
  int32_t pidx[2];
  orc_pidx(g, rank, pidx);
  int invorder[3];
  for (int i = 0; i < 3; ++i) {
    p->order[i] = g->mem_order[axis][i];
    invorder[p->order[i]] = i;
  }
  int j = 0;
  p->size = 1;
  for (int i = 0; i < 3; ++i) {
    int ord = invorder[i];
    int64_t shape;
    if (i != axis) {
      int64_t d = g->gdims_dist[i] / g->pdims[j];
      int64_t mod = g->gdims_dist[i] % g->pdims[j];
      shape = d + ((pidx[j] < mod) ? 1 : 0);
      if (pidx[j] == min64(g->pdims[j], g->gdims_dist[i]) - 1) shape += g->gdims[i] - g->gdims_dist[i];
      p->lo[ord] = (int32_t)(pidx[j] * d + min64(pidx[j], mod));
      j++;
    } else {
      shape = g->gdims[i];
      p->lo[ord] = 0;
    }
    if (shape < 0 || shape > INT32_MAX) return ORC_INVALID_USAGE;
    p->hi[ord] = (int32_t)(p->lo[ord] + shape - 1);
    p->halo_extents[i] = halo_extents ? halo_extents[i] : 0;
    p->padding[i] = padding ? padding[i] : 0;
    if (p->halo_extents[i] < 0 || p->padding[i] < 0) return ORC_INVALID_USAGE;
    int64_t full = shape + 2 * (int64_t)p->halo_extents[i] + p->padding[i];
    if (full < 0 || full > INT32_MAX) return ORC_INVALID_USAGE;
    p->shape[ord] = (int32_t)full;
    if (p->size != 0 && full != 0 && full > INT64_MAX / p->size) return ORC_INVALID_USAGE; /* :492-499 */
    p->size = (p->size == 0 || full == 0) ? 0 : p->size * full;
  }
  return ORC_OK;
}

/* which communicator serves pencil dimension `dim` of an axis-`axis` pencil:
 * first non-axis dim -> column (0), second -> row (1)   (src/cudecomp.cc:1734-1742, halo.h:87-96) */
static int comm_axis_for_dim(int axis, int dim) {
  int count = 0;
  for (int i = 0; i < 3; ++i) {
    if (i == axis) continue;
    if (i == dim) break;
    count++;
  }
  return (count == 0) ? 0 : 1;
}

/* src/cudecomp.cc:1710-1755 */
int orc_shifted_rank(const orc_grid_t* g, int rank, int axis, int dim, int displacement, int periodic,
                     int32_t* shifted_rank) {
  if (axis < 0 || axis > 2 || dim < 0 || dim > 2 || !shifted_rank) return ORC_INVALID_USAGE;
  if (displacement == 0) {
    *shifted_rank = rank;
    return ORC_OK;
  }
  if (dim == axis) {
    *shifted_rank = periodic ? rank : -1;
    return ORC_OK;
  }
  int comm_axis = comm_axis_for_dim(axis, dim);
  int32_t pidx[2];
  orc_pidx(g, rank, pidx);
  int comm_rank = (comm_axis == 0) ? pidx[0] : pidx[1];
  int shifted = comm_rank + displacement;
  if (!periodic && (shifted < 0 || shifted >= g->pdims[comm_axis])) {
    *shifted_rank = -1;
  } else {
    int comm_peer = (shifted + g->pdims[comm_axis]) % g->pdims[comm_axis];
    *shifted_rank = orc_global_rank(g, rank, comm_axis, comm_peer);
  }
  return ORC_OK;
}

/* include/internal/common.h:632-640: round an element count so that count*4 bytes is a
 * multiple of nbytes */
int64_t orc_align_count(int64_t count, int nbytes) {
  int64_t bytes = count * 4;
  int64_t rounded = ((bytes + nbytes - 1) / nbytes) * nbytes;
  return rounded / 4;
}

/* include/internal/common.h:349-366 */
static int64_t global_max_pencil_size(const orc_grid_t* g, int axis) {
  int64_t size = 1;
  int j = 0;
  for (int i = 0; i < 3; ++i) {
    if (i != axis) {
      int64_t dim = (g->gdims_dist[i] + g->pdims[j] - 1) / g->pdims[j];
      dim += g->gdims[i] - g->gdims_dist[i];
      size *= dim;
      j++;
    } else {
      size *= g->gdims[i];
    }
  }
  return size;
}

/* src/cudecomp.cc:1411-1432 */
int64_t orc_transpose_workspace_size(const orc_grid_t* g) {
  int64_t x = global_max_pencil_size(g, 0), y = global_max_pencil_size(g, 1), z = global_max_pencil_size(g, 2);
  int64_t w[4] = {orc_align_count(x, 256) + y, orc_align_count(y, 256) + x, orc_align_count(y, 256) + z,
                  orc_align_count(z, 256) + y};
  int64_t m = w[0];
  for (int i = 1; i < 4; ++i)
    if (w[i] > m) m = w[i];
  return m;
}

static void shape_g(const orc_pinfo_t* p, int64_t out[3]) { /* include/internal/common.h:375-381 */
  for (int i = 0; i < 3; ++i) out[p->order[i]] = p->shape[i];
}

/* src/cudecomp.cc:1434-1459 */
int orc_halo_workspace_size(const orc_grid_t* g, int rank, int axis, const int32_t halo_extents[3], int64_t* size) {
  if (!halo_extents || !size) return ORC_INVALID_USAGE;
  orc_pinfo_t p;
  int rc = orc_pencil_info(g, rank, axis, halo_extents, NULL, &p);
  if (rc) return rc;
  int64_t s[3];
  shape_g(&p, s);
  int64_t hx = 4 * orc_align_count(s[1] * s[2] * p.halo_extents[0], 256);
  int64_t hy = 4 * orc_align_count(s[0] * s[2] * p.halo_extents[1], 256);
  int64_t hz = 4 * orc_align_count(s[0] * s[1] * p.halo_extents[2], 256);
  int64_t m = hx > hy ? hx : hy;
  *size = m > hz ? m : hz;
  return ORC_OK;
}

/* include/internal/common.h:579-589 */
void orc_get_splits(int64_t n, int nchunks, int pad, int64_t* splits) {
  for (int i = 0; i < nchunks; ++i) splits[i] = n / nchunks;
  for (int i = 0; i < n % nchunks; ++i) splits[i] += 1;
  splits[min64(n, nchunks) - 1] += pad;
}

/* include/internal/common.h:533-577.  npergroup: ranks per fast-interconnect group. */
void orc_peer_ranks(int nranks, int npergroup, int rank, int iter, int* src_rank, int* dst_rank) {
  if (nranks == 1 || iter == 0) {
    *src_rank = rank;
    *dst_rank = rank;
    return;
  }
  int power_of_2 = !(nranks & (nranks - 1));
  if (iter % 2 == 1) {
    iter = iter / 2 + 1;
  } else {
    iter = nranks / 2 + iter / 2;
  }
  if (power_of_2) {
    *dst_rank = rank ^ iter;
    *src_rank = rank ^ iter;
  } else {
    int groupid = rank / npergroup;
    if (iter < npergroup) {
      *dst_rank = (rank + iter) % npergroup + groupid * npergroup;
      *src_rank = (rank + npergroup - iter) % npergroup + groupid * npergroup;
    } else {
      int d = (rank + iter) % nranks;
      if (d >= groupid * npergroup && d < (groupid + 1) * npergroup) {
        d += npergroup;
        d %= nranks;
      }
      int s = (rank + nranks - iter) % nranks;
      if (s >= groupid * npergroup && s < (groupid + 1) * npergroup) {
        s += nranks - npergroup;
        s %= nranks;
      }
      *dst_rank = d;
      *src_rank = s;
    }
  }
}

/* include/internal/common.h:369-372: element offset of local coordinate lx (global axis order) */
static int64_t ptr_offset(const orc_pinfo_t* p, const int32_t lx[3]) {
  return (int64_t)lx[p->order[0]] + (int64_t)lx[p->order[1]] * p->shape[0] +
         (int64_t)lx[p->order[2]] * p->shape[0] * p->shape[1];
}

/* include/internal/common.h:620-630 */
static int has_empty_pencils(const orc_grid_t* g, int axis) {
  int j = 0;
  for (int i = 0; i < 3; ++i) {
    if (i != axis) {
      if (g->gdims_dist[i] / g->pdims[j] == 0) return 1;
      j++;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* the two data movers                                                                         */
/* ------------------------------------------------------------------------------------------ */

/* One entry of the batched strided 3-D copy (include/internal/cudecomp_kernels.cuh:125-180):
 * extents [depth,height,width], strides [depth,row], unit column stride, value-preserving. */
static void copy3d(const char* src, char* dst, const int64_t sstr[2], const int64_t dstr[2], const int64_t ext[3],
                   int es) {
  for (int64_t d = 0; d < ext[0]; ++d)
    for (int64_t r = 0; r < ext[1]; ++r)
      memmove(dst + (d * dstr[0] + r * dstr[1]) * es, src + (d * sstr[0] + r * sstr[1]) * es, (size_t)ext[2] * es);
}

/* cutensorPermute as called by localPermute (include/internal/transpose.h:80-157): input modes
 * 0,1,2 with extent_in/strides_in; output mode i is input mode order_out[i] with strides_out[i];
 * alpha = 1, so a pure relocation:  out[sum_i k[order_out[i]]*strides_out[i]] = in[sum_j k[j]*strides_in[j]] */
static void permute3d(const int64_t extent_in[3], const int order_out[3], const int64_t strides_in[3],
                      const int64_t strides_out[3], const char* in, char* out, int es) {
  int64_t so[3]; /* output stride seen from input mode j */
  for (int i = 0; i < 3; ++i) {
    if (extent_in[order_out[i]] == 0) return;
    so[order_out[i]] = strides_out[i];
  }
  for (int64_t k2 = 0; k2 < extent_in[2]; ++k2)
    for (int64_t k1 = 0; k1 < extent_in[1]; ++k1) {
      const char* ip = in + (k1 * strides_in[1] + k2 * strides_in[2]) * es;
      char* op = out + (k1 * so[1] + k2 * so[2]) * es;
      for (int64_t k0 = 0; k0 < extent_in[0]; ++k0)
        memcpy(op + k0 * so[0] * es, ip + k0 * strides_in[0] * es, (size_t)es);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* transpose                                                                                   */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
  orc_pinfo_t a, ah, b, bh;
  int64_t sga[3], sgah[3], sgb[3], sgbh[3];
  char *i1, *o1, *o2, *o3;
  int direct_pack, direct_transpose, done, data_transposed;
  int comm_rank;
  int64_t send_off[ORC_MAX_COMM], recv_off[ORC_MAX_COMM], send_cnt[ORC_MAX_COMM], recv_cnt[ORC_MAX_COMM];
} tctx_t;

typedef struct {
  int ax_a, ax_b, ax_c, comm_axis, P, es, pipelined, orders_equal;
  int32_t in_halo[3], out_halo[3], in_pad[3], out_pad[3];
  int64_t splits_a[ORC_MAX_COMM], splits_b[ORC_MAX_COMM], offsets_a[ORC_MAX_COMM], offsets_b[ORC_MAX_COMM];
} tcfg_t;

/* prod of the (halo-inclusive) extents stored before `axis` in memory (transpose.h:478-482 etc.) */
static int64_t shift_of(const orc_pinfo_t* ph, const int64_t sgh[3], int axis, int64_t offset) {
  int64_t shift = offset;
  for (int i = 0; i < 3; ++i) {
    if (ph->order[i] == axis) break;
    shift *= sgh[ph->order[i]];
  }
  return shift;
}

static void permute_order(const tctx_t* c, int order[3]) { /* transpose.h:434-441 */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      if (c->a.order[j] == c->b.order[i]) {
        order[i] = j;
        break;
      }
}

/* pack phase of one rank: include/internal/transpose.h:425-638 */
static void transpose_pack(const tcfg_t* t, tctx_t* c) {
  const int es = t->es, P = t->P;
  if (c->o1 == c->i1) return; /* skip-pack special case (:630-638) */

  if (c->b.order[2] == t->ax_a && !t->orders_equal) {
    /* Transpose/Pack (:426-531) */
    int64_t extents[3], extents_h[3], extents_h_b[3], strides_in[3] = {1, 0, 0}, strides_out[3] = {1, 0, 0};
    int order[3];
    permute_order(c, order);
    for (int i = 0; i < 3; ++i) {
      extents[i] = c->sga[c->a.order[i]];
      extents_h[i] = c->sgah[c->a.order[i]];
      extents_h_b[i] = c->sgbh[c->a.order[i]];
    }
    for (int i = 1; i < 3; ++i) {
      strides_in[i] = strides_in[i - 1] * extents_h[i - 1];
      strides_out[i] =
          strides_out[i - 1] * (c->direct_transpose ? extents_h_b[order[i - 1]] : extents[order[i - 1]]);
    }
    if (t->pipelined) {
      for (int j = 1; j < P + 1; ++j) {
        int src_rank, dst_rank;
        orc_peer_ranks(P, P, c->comm_rank, j, &src_rank, &dst_rank);
        if (j == P) src_rank = dst_rank = c->comm_rank;
        int64_t shift = shift_of(&c->ah, c->sgah, t->ax_a, t->offsets_a[dst_rank]);
        const char* src = c->i1 + (shift + ptr_offset(&c->ah, t->in_halo)) * es;
        char* dst;
        if (!c->direct_transpose) {
          dst = c->o1 + c->send_off[dst_rank] * es;
        } else {
          int64_t shift_b = shift_of(&c->bh, c->sgbh, t->ax_b, t->offsets_b[src_rank]);
          dst = c->o1 + (shift_b + ptr_offset(&c->bh, t->out_halo)) * es;
        }
        for (int i = 0; i < 3; ++i)
          if (t->ax_a == c->a.order[i]) extents[i] = t->splits_a[dst_rank];
        permute3d(extents, order, strides_in, strides_out, src, dst, es);
      }
    } else {
      const char* src = c->i1 + ptr_offset(&c->ah, t->in_halo) * es;
      char* dst = c->direct_transpose ? c->o1 + ptr_offset(&c->bh, t->out_halo) * es : c->o1;
      permute3d(extents, order, strides_in, strides_out, src, dst, es);
    }
    c->data_transposed = 1;
  } else {
    /* Pack (:533-617): one strided copy per destination, peer-schedule order, self last */
    for (int j = 1; j < P + 1; ++j) {
      int src_rank, dst_rank;
      orc_peer_ranks(P, P, c->comm_rank, j, &src_rank, &dst_rank);
      if (j == P) src_rank = dst_rank = c->comm_rank;
      int64_t shift = shift_of(&c->ah, c->sgah, t->ax_a, t->offsets_a[dst_rank]);
      const char* src = c->i1 + (shift + ptr_offset(&c->ah, t->in_halo)) * es;
      char* dst;
      int64_t sstr[2], dstr[2], ext[3];
      sstr[0] = (int64_t)c->ah.shape[0] * c->ah.shape[1];
      sstr[1] = c->ah.shape[0];
      if (!c->direct_pack) {
        dst = c->o1 + c->send_off[dst_rank] * es;
        dstr[1] = (t->ax_a == c->a.order[0]) ? t->splits_a[dst_rank] : c->a.shape[0];
        dstr[0] = dstr[1] * ((t->ax_a == c->a.order[1]) ? t->splits_a[dst_rank] : c->a.shape[1]);
      } else {
        int64_t shift_b = shift_of(&c->bh, c->sgbh, t->ax_b, t->offsets_b[src_rank]);
        dst = c->o1 + (shift_b + ptr_offset(&c->bh, t->out_halo)) * es;
        dstr[0] = (int64_t)c->bh.shape[0] * c->bh.shape[1];
        dstr[1] = c->bh.shape[0];
      }
      ext[2] = (t->ax_a == c->a.order[0]) ? t->splits_a[dst_rank] : c->a.shape[0];
      ext[1] = (t->ax_a == c->a.order[1]) ? t->splits_a[dst_rank] : c->a.shape[1];
      ext[0] = (t->ax_a == c->a.order[2]) ? t->splits_a[dst_rank] : c->a.shape[2];
      copy3d(src, dst, sstr, dstr, ext, es);
    }
  }
  if (c->o1 == c->o3) c->done = 1; /* :619-629 */
}

/* unpack phase of one rank: include/internal/transpose.h:650-895 */
static void transpose_unpack(const tcfg_t* t, tctx_t* c) {
  const int es = t->es, P = t->P;
  if (c->done) return;
  if (P == 1) c->o2 = c->o1; /* :646-648 */

  if (!c->data_transposed && !t->orders_equal) {
    int order[3];
    permute_order(c, order);
    if (c->a.order[2] == t->ax_b || P == 1) {
      /* Transpose/Unpack (:652-756) */
      int64_t extents[3], extents_h[3], extents_h_a[3], strides_in[3] = {1, 0, 0}, strides_out[3] = {1, 0, 0};
      for (int i = 0; i < 3; ++i) {
        extents[i] = c->sgb[c->a.order[i]];
        extents_h[i] = c->sgbh[c->bh.order[i]];
        extents_h_a[i] = c->sgah[c->ah.order[i]];
        if (i > 0) {
          strides_in[i] = strides_in[i - 1] * (c->direct_transpose ? extents_h_a[i - 1] : extents[i - 1]);
          strides_out[i] = strides_out[i - 1] * extents_h[i - 1];
        }
      }
      if (t->pipelined) {
        for (int j = 0; j < P; ++j) {
          int src_rank, dst_rank;
          orc_peer_ranks(P, P, c->comm_rank, j, &src_rank, &dst_rank);
          if (j == 0) src_rank = dst_rank = c->comm_rank;
          if (c->o2 != c->o3) {
            int64_t shift = shift_of(&c->bh, c->sgbh, t->ax_b, t->offsets_b[src_rank]);
            const char* src;
            if (!c->direct_transpose) {
              src = c->o2 + c->recv_off[src_rank] * es;
            } else {
              int64_t shift_a = shift_of(&c->ah, c->sgah, t->ax_a, t->offsets_a[dst_rank]);
              src = c->o2 + (shift_a + ptr_offset(&c->ah, t->in_halo)) * es;
            }
            char* dst = c->o3 + (shift + ptr_offset(&c->bh, t->out_halo)) * es;
            for (int i = 0; i < 3; ++i)
              if (t->ax_b == c->a.order[i]) {
                extents[i] = t->splits_b[src_rank];
                break;
              }
            permute3d(extents, order, strides_in, strides_out, src, dst, es);
          }
        }
      } else if (c->o2 != c->o3) {
        const char* src = c->direct_transpose ? c->o2 + ptr_offset(&c->ah, t->in_halo) * es : c->o2;
        char* dst = c->o3 + ptr_offset(&c->bh, t->out_halo) * es;
        permute3d(extents, order, strides_in, strides_out, src, dst, es);
      }
    } else {
      /* Split Transpose/Unpack (:757-828): one permute per source chunk */
      int64_t extents[3], extents_h[3], strides_in[3] = {1, 0, 0}, strides_out[3] = {1, 0, 0};
      for (int i = 0; i < 3; ++i) {
        extents[i] = c->sgb[c->a.order[i]];
        extents_h[i] = c->sgbh[c->bh.order[i]];
        if (i > 0) strides_out[i] = strides_out[i - 1] * extents_h[i - 1];
      }
      for (int j = 0; j < P; ++j) {
        int src_rank, dst_rank;
        orc_peer_ranks(P, P, c->comm_rank, j, &src_rank, &dst_rank);
        if (j == 0) src_rank = dst_rank = c->comm_rank;
        if (c->o2 != c->o3) {
          for (int i = 0; i < 3; ++i) {
            if (t->ax_b == c->a.order[i]) extents[i] = t->splits_b[src_rank];
            if (i > 0) strides_in[i] = strides_in[i - 1] * extents[i - 1];
          }
          int64_t shift = shift_of(&c->bh, c->sgbh, t->ax_b, t->offsets_b[src_rank]);
          const char* src = c->o2 + c->recv_off[src_rank] * es;
          char* dst = c->o3 + (shift + ptr_offset(&c->bh, t->out_halo)) * es;
          permute3d(extents, order, strides_in, strides_out, src, dst, es);
        }
      }
    }
  } else {
    /* Unpack (:830-895): one strided copy per source chunk, self first */
    for (int j = 0; j < P; ++j) {
      int src_rank, dst_rank;
      orc_peer_ranks(P, P, c->comm_rank, j, &src_rank, &dst_rank);
      if (j == 0) src_rank = dst_rank = c->comm_rank;
      if (c->o2 != c->o3) {
        int64_t shift = shift_of(&c->bh, c->sgbh, t->ax_b, t->offsets_b[src_rank]);
        const char* src = c->o2 + c->recv_off[src_rank] * es;
        char* dst = c->o3 + (shift + ptr_offset(&c->bh, t->out_halo)) * es;
        int64_t sstr[2], dstr[2], ext[3];
        sstr[1] = (t->ax_b == c->b.order[0]) ? t->splits_b[src_rank] : c->b.shape[0];
        sstr[0] = sstr[1] * ((t->ax_b == c->b.order[1]) ? t->splits_b[src_rank] : c->b.shape[1]);
        dstr[0] = (int64_t)c->bh.shape[0] * c->bh.shape[1];
        dstr[1] = c->bh.shape[0];
        ext[2] = (t->ax_b == c->b.order[0]) ? t->splits_b[src_rank] : c->b.shape[0];
        ext[1] = (t->ax_b == c->b.order[1]) ? t->splits_b[src_rank] : c->b.shape[1];
        ext[0] = (t->ax_b == c->b.order[2]) ? t->splits_b[src_rank] : c->b.shape[2];
        copy3d(src, dst, sstr, dstr, ext, es);
      }
    }
  }
}

/* configuration shared by all ranks of one transpose call (transpose.h:222-259) */
static int transpose_setup(const orc_grid_t* g, int ax, int dir, int es, const int32_t in_halo[3],
                           const int32_t out_halo[3], const int32_t in_pad[3], const int32_t out_pad[3], int pipelined,
                           tcfg_t* tp, int flags[3]) {
  if (es != 4 && es != 8 && es != 16) return ORC_INVALID_USAGE;
  if (ax < 0 || ax > 2 || dir == 0) return ORC_INVALID_USAGE;
  tcfg_t t;
  memset(&t, 0, sizeof(t));
  t.es = es;
  t.pipelined = pipelined;
  for (int i = 0; i < 3; ++i) {
    t.in_halo[i] = in_halo ? in_halo[i] : 0;
    t.out_halo[i] = out_halo ? out_halo[i] : 0;
    t.in_pad[i] = in_pad ? in_pad[i] : 0;
    t.out_pad[i] = out_pad ? out_pad[i] : 0;
  }
  int input_hp = 0, output_hp = 0, hp_equal = 1;
  for (int i = 0; i < 3; ++i) {
    if (t.in_halo[i] || t.in_pad[i]) input_hp = 1;
    if (t.out_halo[i] || t.out_pad[i]) output_hp = 1;
    if (t.in_halo[i] != t.out_halo[i] || t.in_pad[i] != t.out_pad[i]) hp_equal = 0;
  }
  flags[0] = input_hp;
  flags[1] = output_hp;
  flags[2] = hp_equal;

  /* axes and communicator (transpose.h:222-245) */
  int fwd = dir > 0;
  t.ax_a = ax;
  t.ax_b = (fwd ? ax + 1 : ax + 2) % 3;
  t.ax_c = (fwd ? ax + 2 : ax + 1) % 3;
  t.comm_axis = (t.ax_a == 2 || t.ax_b == 2) ? 1 : 0;
  t.P = g->pdims[t.comm_axis == 0 ? 0 : 1];
  orc_get_splits(g->gdims_dist[t.ax_a], t.P, g->gdims[t.ax_a] - g->gdims_dist[t.ax_a], t.splits_a);
  orc_get_splits(g->gdims_dist[t.ax_b], t.P, g->gdims[t.ax_b] - g->gdims_dist[t.ax_b], t.splits_b);
  for (int i = 0; i + 1 < t.P; ++i) {
    t.offsets_a[i + 1] = t.offsets_a[i] + t.splits_a[i];
    t.offsets_b[i + 1] = t.offsets_b[i] + t.splits_b[i];
  }
  if (has_empty_pencils(g, t.ax_a) || has_empty_pencils(g, t.ax_b)) return ORC_NOT_SUPPORTED; /* :257-259 */
  /* the two memory orders are the same on every rank */
  orc_pinfo_t a0, b0;
  int rc;
  if ((rc = orc_pencil_info(g, 0, t.ax_a, NULL, NULL, &a0))) return rc;
  if ((rc = orc_pencil_info(g, 0, t.ax_b, NULL, NULL, &b0))) return rc;
  t.orders_equal = 1;
  for (int i = 0; i < 3; ++i)
    if (a0.order[i] != b0.order[i]) t.orders_equal = 0;
  *tp = t;
  return ORC_OK;
}

/* one rank's pencils, phase pointers, special cases and exchange counts (transpose.h:281-421) */
static int transpose_rank_context(const orc_grid_t* g, const tcfg_t* tp, const int flags[3], int r, void* in, void* out,
                                  void* work, tctx_t* c) {
  const tcfg_t t = *tp;
  const int input_hp = flags[0], output_hp = flags[1], hp_equal = flags[2], es = t.es;
  int rc;
  if ((rc = orc_pencil_info(g, r, t.ax_a, NULL, NULL, &c->a))) return rc;
  if ((rc = orc_pencil_info(g, r, t.ax_a, t.in_halo, t.in_pad, &c->ah))) return rc;
  if ((rc = orc_pencil_info(g, r, t.ax_b, NULL, NULL, &c->b))) return rc;
  if ((rc = orc_pencil_info(g, r, t.ax_b, t.out_halo, t.out_pad, &c->bh))) return rc;
  shape_g(&c->a, c->sga);
  shape_g(&c->ah, c->sgah);
  shape_g(&c->b, c->sgb);
  shape_g(&c->bh, c->sgbh);
  int32_t pidx[2];
  orc_pidx(g, r, pidx);
  c->comm_rank = (t.comm_axis == 0) ? pidx[0] : pidx[1];
  int inplace = (in == out);
  /* phase pointers (transpose.h:281-284) */
  c->i1 = (char*)in;
  c->o1 = (char*)work;
  c->o2 = (char*)work + orc_align_count(c->a.size, 256) * es;
  c->o3 = (char*)out;
  /* special cases (transpose.h:323-404) */
  if (t.P == 1) {
    if (t.orders_equal) {
      if (inplace) {
        if (hp_equal) c->done = 1; /* nothing to do */
      } else {
        c->o1 = c->o3;
        c->direct_pack = 1;
      }
    } else if (!inplace) {
      if (c->b.order[2] == t.ax_a) {
        c->o1 = c->o3;
        c->direct_transpose = 1;
      } else {
        c->o1 = c->i1;
        c->o2 = c->i1;
        c->direct_transpose = 1;
      }
    }
  } else {
    int enable = !(t.pipelined && inplace);
    if (enable) {
      if (c->a.order[2] == t.ax_a && !input_hp) {
        c->o1 = c->i1;
        c->o2 = (char*)work;
      } else if (c->a.order[2] == t.ax_b && t.orders_equal && !output_hp) {
        c->o2 = c->o3;
      }
    }
  }
  /* counts / offsets (transpose.h:407-421) */
  for (int i = 0; i < t.P; ++i) {
    c->send_off[i] = t.offsets_a[i] * c->sga[t.ax_b] * c->sga[t.ax_c];
    c->recv_off[i] = t.offsets_b[i] * c->sgb[t.ax_a] * c->sgb[t.ax_c];
    c->send_cnt[i] = t.splits_a[i] * c->sga[t.ax_b] * c->sga[t.ax_c];
    c->recv_cnt[i] = t.splits_b[i] * c->sgb[t.ax_a] * c->sgb[t.ax_c];
  }
  return ORC_OK;
}

int orc_transpose(const orc_grid_t* g, int ax, int dir, int es, void* const* in, void* const* out, void* const* work,
                  const int32_t in_halo[3], const int32_t out_halo[3], const int32_t in_pad[3],
                  const int32_t out_pad[3], int pipelined) {
  tcfg_t t;
  int flags[3];
  int rc = transpose_setup(g, ax, dir, es, in_halo, out_halo, in_pad, out_pad, pipelined, &t, flags);
  if (rc != ORC_OK) return rc;
  const int nranks = orc_nranks(g);
  tctx_t* ctx = (tctx_t*)calloc((size_t)nranks, sizeof(tctx_t));
  if (!ctx) return ORC_NOT_SUPPORTED;
  for (int r = 0; r < nranks && rc == ORC_OK; ++r)
    rc = transpose_rank_context(g, &t, flags, r, in[r], out[r], work[r], &ctx[r]);

  if (rc == ORC_OK) {
    for (int r = 0; r < nranks; ++r)
      if (!ctx[r].done) transpose_pack(&t, &ctx[r]);

    /* all-to-all inside each row/column communicator: chunk d of rank r's send buffer lands at
     * slot r of rank d's receive buffer (MPI_Alltoallv semantics, comm_routines.h:363-413) */
    if (t.P > 1) {
      for (int r = 0; r < nranks; ++r) {
        tctx_t* c = &ctx[r];
        for (int d = 0; d < t.P; ++d) {
          int peer = orc_global_rank(g, r, t.comm_axis, d);
          tctx_t* pc = &ctx[peer];
          if (c->send_cnt[d] != pc->recv_cnt[c->comm_rank]) {
            rc = ORC_NOT_SUPPORTED; /* would be a plan bug */
            break;
          }
          memcpy(pc->o2 + pc->recv_off[c->comm_rank] * es, c->o1 + c->send_off[d] * es, (size_t)c->send_cnt[d] * es);
        }
      }
    }
    if (rc == ORC_OK)
      for (int r = 0; r < nranks; ++r) transpose_unpack(&t, &ctx[r]);
  }
  free(ctx);
  return rc;
}

/* One rank of a transpose with the exchange delegated to the caller (e.g. MPI_Alltoallv over host memory: the
 * "host-MPI CPU path" used as the CPU baseline, oracle/cpu_mpi_cycle.c).  Same pack / unpack code as above. */
int orc_transpose_rank(const orc_grid_t* g, int rank, int ax, int dir, int es, void* in, void* out, void* work,
                       const int32_t in_halo[3], const int32_t out_halo[3], const int32_t in_pad[3],
                       const int32_t out_pad[3], int pipelined, orc_exchange_fn exchange, void* user) {
  tcfg_t t;
  int flags[3];
  int rc = transpose_setup(g, ax, dir, es, in_halo, out_halo, in_pad, out_pad, pipelined, &t, flags);
  if (rc != ORC_OK) return rc;
  if (rank < 0 || rank >= orc_nranks(g)) return ORC_INVALID_USAGE;
  tctx_t c;
  memset(&c, 0, sizeof(c));
  if ((rc = transpose_rank_context(g, &t, flags, rank, in, out, work, &c))) return rc;
  if (c.done) return ORC_OK;
  transpose_pack(&t, &c);
  if (t.P > 1) {
    if (!exchange) return ORC_INVALID_USAGE;
    exchange(user, c.o1, c.send_cnt, c.send_off, c.o2, c.recv_cnt, c.recv_off, t.P, t.comm_axis, c.comm_rank, es);
  }
  transpose_unpack(&t, &c);
  return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* halo update                                                                                 */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
  orc_pinfo_t h, hp;
  int64_t sgh[3], sghp[3];
  int32_t nb[2];
  int c;
} hctx_t;

/* K1 descriptors shared by the three halo cases: extents come from the un-padded pencil, strides
 * from the padded one (halo.h:181-189, 215-224, 252-261) */
static void halo_extents(const hctx_t* c, int dim, int32_t he, int64_t ext[3]) {
  ext[0] = (dim == c->h.order[2]) ? he : c->h.shape[2];
  ext[1] = (dim == c->h.order[1]) ? he : c->h.shape[1];
  ext[2] = (dim == c->h.order[0]) ? he : c->h.shape[0];
}

int orc_update_halos(const orc_grid_t* g, int ax, int es, void* const* data, void* const* work,
                     const int32_t halo_extents_in[3], const int32_t periods_in[3], int dim,
                     const int32_t padding_in[3], int staged) {
  if (es != 4 && es != 8 && es != 16) return ORC_INVALID_USAGE;
  if (ax < 0 || ax > 2 || dim < 0 || dim > 2 || !halo_extents_in) return ORC_INVALID_USAGE;
  int32_t he[3], per[3], pad[3];
  for (int i = 0; i < 3; ++i) {
    he[i] = halo_extents_in[i];
    per[i] = periods_in ? periods_in[i] : 0;
    pad[i] = padding_in ? padding_in[i] : 0;
  }
  if (he[0] == 0 && he[1] == 0 && he[2] == 0) return ORC_OK; /* src/cudecomp.cc:1930-1933 */
  if (has_empty_pencils(g, ax)) return ORC_NOT_SUPPORTED;      /* halo.h:57-59 */
  if (he[dim] == 0) return ORC_OK;                              /* halo.h:71 */

  const int nranks = orc_nranks(g);
  const int comm_axis = comm_axis_for_dim(ax, dim);
  const int P = g->pdims[comm_axis];
  int64_t splits[ORC_MAX_COMM];
  int has_pad = pad[0] || pad[1] || pad[2];

  hctx_t* ctx = (hctx_t*)calloc((size_t)nranks, sizeof(hctx_t));
  if (!ctx) return ORC_NOT_SUPPORTED;
  int rc = ORC_OK;
  for (int r = 0; r < nranks && rc == ORC_OK; ++r) {
    hctx_t* c = &ctx[r];
    if ((rc = orc_pencil_info(g, r, ax, he, NULL, &c->h))) break;
    if ((rc = orc_pencil_info(g, r, ax, he, pad, &c->hp))) break;
    shape_g(&c->h, c->sgh);
    shape_g(&c->hp, c->sghp);
    orc_shifted_rank(g, r, ax, dim, -1, per[dim], &c->nb[0]);
    orc_shifted_rank(g, r, ax, dim, 1, per[dim], &c->nb[1]);
    /* case selection (halo.h:98-162) */
    c->c = (dim != c->h.order[0] && dim != c->h.order[1]) ? 2 : 1;
    if (c->nb[0] == r && c->nb[1] == r) {
      c->c = 0;
    } else if (c->nb[0] == -1 && c->nb[1] == -1) {
      c->c = -1; /* nothing to do */
    } else {
      /* nearest-neighbour only (halo.h:120-144) */
      orc_get_splits(g->gdims_dist[dim], P, g->gdims[dim] - g->gdims_dist[dim], splits);
      int32_t pidx[2];
      orc_pidx(g, r, pidx);
      int cr = (comm_axis == 0) ? pidx[0] : pidx[1];
      int l = cr - 1, rr = cr + 1;
      if (per[dim]) {
        l = (l + P) % P;
        rr = (rr + P) % P;
      }
      if (l >= 0 && (he[dim] > splits[l] || he[dim] > splits[cr])) rc = ORC_INVALID_USAGE;
      if (rr < P && (he[dim] > splits[rr] || he[dim] > splits[cr])) rc = ORC_INVALID_USAGE;
    }
    if (c->c == 2 && (has_pad || staged)) c->c = 1;
  }

  /* 1) pack (c==1) / self-copy (c==0) on every rank */
  for (int r = 0; r < nranks && rc == ORC_OK; ++r) {
    hctx_t* c = &ctx[r];
    char* buf = (char*)data[r];
    int64_t ext[3], str[2] = {(int64_t)c->hp.shape[0] * c->hp.shape[1], c->hp.shape[0]};
    int32_t lx[3] = {0, 0, 0}, zero[3] = {0, 0, 0};
    halo_extents(c, dim, he[dim], ext);
    if (c->c == 0) {
      /* periodic self-copy (halo.h:165-193) */
      lx[dim] = (int32_t)(c->sghp[dim] - 2 * he[dim] - pad[dim]);
      copy3d(buf + ptr_offset(&c->hp, lx) * es, buf + ptr_offset(&c->hp, zero) * es, str, str, ext, es);
      lx[dim] = he[dim];
      const char* src = buf + ptr_offset(&c->hp, lx) * es;
      lx[dim] = (int32_t)(c->sghp[dim] - he[dim] - pad[dim]);
      copy3d(src, buf + ptr_offset(&c->hp, lx) * es, str, str, ext, es);
    } else if (c->c == 1) {
      /* pack both faces into work[0], work[A] (halo.h:195-227) */
      int64_t halo_size = c->sgh[(dim + 1) % 3] * c->sgh[(dim + 2) % 3] * he[dim];
      int64_t A = orc_align_count(halo_size, 256);
      int64_t dstr[2];
      dstr[1] = (dim == c->h.order[0]) ? he[dim] : c->h.shape[0];
      dstr[0] = dstr[1] * ((dim == c->h.order[1]) ? he[dim] : c->h.shape[1]);
      char* send = (char*)work[r];
      lx[dim] = he[dim];
      copy3d(buf + ptr_offset(&c->hp, lx) * es, send, str, dstr, ext, es);
      lx[dim] = (int32_t)(c->sghp[dim] - 2 * he[dim] - pad[dim]);
      copy3d(buf + ptr_offset(&c->hp, lx) * es, send + A * es, str, dstr, ext, es);
    }
  }

  /* 2) neighbour exchange (comm_routines.h:686-707): face (i+1)%2 goes to neighbour (i+1)%2,
   *    slot i is filled by neighbour i.  Receiver-driven here: slot 0 (left halo) takes the left
   *    neighbour's right face (its send slot 1), slot 1 takes the right neighbour's left face. */
  for (int r = 0; r < nranks && rc == ORC_OK; ++r) {
    hctx_t* c = &ctx[r];
    if (c->c != 1 && c->c != 2) continue;
    for (int i = 0; i < 2; ++i) {
      int peer = c->nb[i];
      if (peer == -1) continue;
      hctx_t* pc = &ctx[peer];
      int64_t halo_size = c->sgh[(dim + 1) % 3] * c->sgh[(dim + 2) % 3] * he[dim];
      int64_t A = orc_align_count(halo_size, 256);
      int32_t lx[3] = {0, 0, 0};
      const char* src;
      char* dst;
      if (pc->c == 1) {
        src = (const char*)work[peer] + (i == 0 ? A : 0) * es;
      } else { /* contiguous face inside the peer's pencil (halo.h:287-295) */
        lx[dim] = (i == 0) ? (int32_t)(pc->sghp[dim] - 2 * he[dim]) : he[dim];
        src = (const char*)data[peer] + ptr_offset(&pc->h, lx) * es;
      }
      if (c->c == 1) {
        dst = (char*)work[r] + (2 * A + (i == 0 ? 0 : A)) * es;
      } else {
        lx[dim] = (i == 0) ? 0 : (int32_t)(c->sghp[dim] - he[dim]);
        dst = (char*)data[r] + ptr_offset(&c->h, lx) * es;
      }
      memcpy(dst, src, (size_t)halo_size * es);
    }
  }

  /* 3) unpack (halo.h:242-276) */
  for (int r = 0; r < nranks && rc == ORC_OK; ++r) {
    hctx_t* c = &ctx[r];
    if (c->c != 1) continue;
    char* buf = (char*)data[r];
    int64_t ext[3], dstr[2] = {(int64_t)c->hp.shape[0] * c->hp.shape[1], c->hp.shape[0]}, sstr[2];
    halo_extents(c, dim, he[dim], ext);
    int64_t halo_size = c->sgh[(dim + 1) % 3] * c->sgh[(dim + 2) % 3] * he[dim];
    int64_t A = orc_align_count(halo_size, 256);
    sstr[1] = (dim == c->h.order[0]) ? he[dim] : c->h.shape[0];
    sstr[0] = sstr[1] * ((dim == c->h.order[1]) ? he[dim] : c->h.shape[1]);
    const char* recv = (const char*)work[r] + 2 * A * es;
    int32_t lx[3] = {0, 0, 0}, zero[3] = {0, 0, 0};
    if (c->nb[0] != -1) copy3d(recv, buf + ptr_offset(&c->hp, zero) * es, sstr, dstr, ext, es);
    if (c->nb[1] != -1) {
      lx[dim] = (int32_t)(c->sghp[dim] - he[dim] - pad[dim]);
      copy3d(recv + A * es, buf + ptr_offset(&c->hp, lx) * es, sstr, dstr, ext, es);
    }
  }
  free(ctx);
  return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* the reference's analytic test oracle                                                        */
/* ------------------------------------------------------------------------------------------ */

static void put_value(void* data, int64_t i, int kind, double re, double im) {
  switch (kind) {
  case 0: ((float*)data)[i] = (float)re; break;
  case 1: ((double*)data)[i] = re; break;
  case 2:
    ((float*)data)[2 * i] = (float)re;
    ((float*)data)[2 * i + 1] = (float)im;
    break;
  default:
    ((double*)data)[2 * i] = re;
    ((double*)data)[2 * i + 1] = im;
    break;
  }
}

static int is_internal(const orc_pinfo_t* p, const int64_t l[3]) { /* tests/ctest/transpose_tests.cc:312-319 */
  for (int i = 0; i < 3; ++i) {
    int o = p->order[i];
    if (l[i] < p->halo_extents[o] || l[i] >= p->shape[i] - p->halo_extents[o] - p->padding[o]) return 0;
  }
  return 1;
}

/* tests/ctest/transpose_tests.cc:333-354 (halo_style 0: unset complex = (-1,-1)) and
 * tests/ctest/halo_tests.cc:214-227 (halo_style 1: unset complex = (-1,0)) */
void orc_fill_pencil(const orc_pinfo_t* p, const int32_t gdims[3], int kind, int halo_style, void* data) {
  for (int64_t i = 0; i < p->size; ++i) {
    int64_t l[3] = {i % p->shape[0], i / p->shape[0] % p->shape[1], i / ((int64_t)p->shape[0] * p->shape[1])};
    if (!is_internal(p, l)) {
      put_value(data, i, kind, -1.0, halo_style ? 0.0 : -1.0);
      continue;
    }
    int64_t gl[3];
    for (int k = 0; k < 3; ++k) gl[p->order[k]] = l[k] + p->lo[k] - p->halo_extents[p->order[k]];
    double v = (double)(gl[0] + gdims[0] * (gl[1] + gl[2] * (int64_t)gdims[1]));
    put_value(data, i, kind, v, -v);
  }
}

/* tests/ctest/halo_tests.cc:229-253 */
void orc_fill_halo_reference(const orc_pinfo_t* p, const int32_t gdims[3], const int32_t periods[3], int kind,
                             void* data) {
  for (int64_t i = 0; i < p->size; ++i) {
    int64_t l[3] = {i % p->shape[0], i / p->shape[0] % p->shape[1], i / ((int64_t)p->shape[0] * p->shape[1])};
    int64_t gl[3];
    int unset = 0;
    for (int k = 0; k < 3; ++k) {
      gl[p->order[k]] = l[k] + p->lo[k] - p->halo_extents[p->order[k]];
      if (l[k] >= p->shape[k] - p->padding[p->order[k]]) unset = 1;
    }
    for (int d = 0; d < 3; ++d) {
      if (gl[d] >= 0 && gl[d] < gdims[d]) continue;
      if (periods[d]) {
        int64_t w = gl[d] % gdims[d];
        gl[d] = (w < 0) ? w + gdims[d] : w;
      } else {
        unset = 1;
      }
    }
    if (unset) {
      put_value(data, i, kind, -1.0, 0.0);
    } else {
      double v = (double)(gl[0] + gdims[0] * (gl[1] + gl[2] * (int64_t)gdims[1]));
      put_value(data, i, kind, v, -v);
    }
  }
}

/* tests/ctest/transpose_tests.cc:356-378 (interior only) / halo_tests.cc:255-272 (whole buffer) */
int64_t orc_compare_pencil(const orc_pinfo_t* p, int es, const void* expected, const void* actual,
                           int interior_only) {
  const char* e = (const char*)expected;
  const char* a = (const char*)actual;
  for (int64_t i = 0; i < p->size; ++i) {
    if (memcmp(e + i * es, a + i * es, (size_t)es) == 0) continue;
    if (interior_only) {
      int64_t l[3] = {i % p->shape[0], i / p->shape[0] % p->shape[1], i / ((int64_t)p->shape[0] * p->shape[1])};
      if (!is_internal(p, l)) continue;
    }
    return i + 1;
  }
  return 0;
}
