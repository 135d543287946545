/*
 * oracle/ref_ops.c  --  CPU restatement of the reference's message-passing arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under pgl_amd/ links, loads or calls this file.
 * It is the checker for tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg ("kind": "port"), never the thing shipped or measured as the product.
 *
 * What it restates.  PGL (reference @ /root/reference, pgl 2.2.6) owns no arithmetic on
 * this path: every floating point operation is a PaddlePaddle op reached from
 *     pgl/graph.py:859-861,885-887   paddle.geometric.send_u_recv
 *     pgl/graph.py:929-937           paddle.geometric.send_ue_recv
 *     pgl/graph.py:964-966           paddle.geometric.send_uv
 *     pgl/math.py:36-42,79,113,145,178   paddle.geometric.segment_{sum,mean,max,min}
 *     pgl/math.py:216-224            segment_softmax (composed: max, gather, sub, exp, sum, gather, div)
 *     pgl/utils/helper.py:156-160    paddle.unique(return_inverse=True)
 *     pgl/graph_kernel.pyx:59-88     build_index (the only arithmetic-free native code PGL owns)
 * PaddlePaddle (pin ">= 2.2.0", README.md:146; paddle.geometric.* implies 2.4.x-2.5.x) is an
 * un-vendored third-party dependency, absent from /root/reference and from this image.  The
 * functions below restate the published CPU algorithm of those ops (Paddle phi CPU kernels
 * send_u_recv / send_ue_recv / send_uv / segment_pool):
 *   - output zero-filled first; rows that receive nothing stay 0 (NOT +-inf) for every reduce op;
 *   - SUM: one serial pass over the edges IN RAW COO ORDER, out[dst[e]] += x[src[e]];
 *   - MEAN: SUM, then rows with count>0 are divided by their in-count;
 *   - MAX/MIN: first message copied into the row, later ones combined elementwise;
 *   - out rows M = out_size if out_size > 0 else x.shape[0];
 *   - segment_*: ids sorted non-decreasing, out rows = ids[last]+1, missing ids -> 0,
 *     mean divides by occurrence count.
 *
 * Pinning.  The restatement is pinned by the reference's own golden vectors (SURVEY.md
 * Appendix B, G1-G10: tests/test_graph.py:337-410, tests/test_dist_graph.py:115-137,
 * tests/test_math.py:32-66, tests/test_graph_op.py:56-68, pgl/math.py docstrings) in
 * tests/test_oracle_golden.py, and ref_build_index is additionally checked bit-for-bit against
 * the reference's own compiled graph_kernel.pyx (oracle/_ref).  mean/max/min of send_u_recv,
 * send_uv, and send_ue_recv with mul are NOT pinned by any reference test ("parity unpinned"
 * for those; they are cross-checked against independent scipy/torch-CPU formulations instead).
 *
 * Build: gcc -O3 -fPIC -shared -fopenmp oracle/ref_ops.c -o oracle/_build/libref_ops.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { R_SUM = 0, R_MEAN = 1, R_MAX = 2, R_MIN = 3 };
enum { M_ADD = 0, M_SUB = 1, M_MUL = 2, M_DIV = 3 };

/* ------------------------------------------------------------------------------------------
 * send_u_recv  (graph.py:859-861 -> paddle.geometric.send_u_recv; CPU: serial edge loop)
 *   x [n_in, d], src/dst [E] int64 in raw edge order, out [m, d] (fully written here).
 *   cnt: caller scratch int64[m] (used by MEAN/MAX/MIN), may be NULL for SUM.
 * ---------------------------------------------------------------------------------------- */
#define DEF_SEND_U_RECV(NAME, T)                                                              \
    void NAME(const T* x, const int64_t* src, const int64_t* dst, int64_t E, int64_t m,       \
              int64_t d, int op, T* out, int64_t* cnt) {                                      \
        memset(out, 0, sizeof(T) * (size_t)m * (size_t)d);                                    \
        if (cnt) memset(cnt, 0, sizeof(int64_t) * (size_t)m);                                 \
        for (int64_t e = 0; e < E; ++e) {                                                     \
            const T* xs = x + src[e] * d;                                                     \
            T* od = out + dst[e] * d;                                                         \
            if (op == R_SUM || op == R_MEAN) {                                                \
                for (int64_t j = 0; j < d; ++j) od[j] += xs[j];                               \
            } else if (cnt[dst[e]] == 0) {                                                    \
                for (int64_t j = 0; j < d; ++j) od[j] = xs[j];                                \
            } else if (op == R_MAX) {                                                         \
                for (int64_t j = 0; j < d; ++j) od[j] = xs[j] > od[j] ? xs[j] : od[j];        \
            } else {                                                                          \
                for (int64_t j = 0; j < d; ++j) od[j] = xs[j] < od[j] ? xs[j] : od[j];        \
            }                                                                                 \
            if (cnt) cnt[dst[e]] += 1;                                                        \
        }                                                                                     \
        if (op == R_MEAN) {                                                                   \
            for (int64_t r = 0; r < m; ++r) {                                                 \
                if (cnt[r] == 0) continue;                                                    \
                for (int64_t j = 0; j < d; ++j) out[r * d + j] = out[r * d + j] / (T)cnt[r];  \
            }                                                                                 \
        }                                                                                     \
    }

DEF_SEND_U_RECV(ref_send_u_recv_f32, float)
DEF_SEND_U_RECV(ref_send_u_recv_f64, double)
DEF_SEND_U_RECV(ref_send_u_recv_i64, int64_t)
DEF_SEND_U_RECV(ref_send_u_recv_i32, int32_t)

/* OpenMP row-parallel CSR variant of SUM: NOT the reference's
 * algorithm, only the "all host cores" comparison point next to the serial port.
 *   indptr [m+1], col [E] (dst-sorted CSR, as build_index emits with u=dst, v=src). */
void ref_csr_spmm_sum_f32_omp(const float* x, const int64_t* indptr, const int64_t* col,
                              int64_t m, int64_t d, float* out) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t r = 0; r < m; ++r) {
        float* o = out + r * d;
        for (int64_t j = 0; j < d; ++j) o[j] = 0.f;
        for (int64_t p = indptr[r]; p < indptr[r + 1]; ++p) {
            const float* xs = x + col[p] * d;
            for (int64_t j = 0; j < d; ++j) o[j] += xs[j];
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * send_ue_recv (graph.py:929-937).  message m[e, j] = x[src[e], xi[j]] (mop) y[e, yi[j]],
 * j in [0, dout): numpy broadcasting between the trailing dims of x[src] and y is flattened
 * by the caller into the two index maps xi/yi (len dout).  Reduce exactly as send_u_recv.
 * ---------------------------------------------------------------------------------------- */
#define APPLY_MOP(T, a, b, mop) \
    ((mop) == M_ADD ? (T)((a) + (b)) : (mop) == M_SUB ? (T)((a) - (b)) : (mop) == M_MUL ? (T)((a) * (b)) : (T)((a) / (b)))

#define DEF_SEND_UE_RECV(NAME, T)                                                             \
    void NAME(const T* x, const T* y, const int64_t* src, const int64_t* dst, int64_t E,      \
              int64_t m, int64_t dx, int64_t dy, int64_t dout, const int32_t* xi,             \
              const int32_t* yi, int mop, int rop, T* out, int64_t* cnt) {                    \
        memset(out, 0, sizeof(T) * (size_t)m * (size_t)dout);                                 \
        if (cnt) memset(cnt, 0, sizeof(int64_t) * (size_t)m);                                 \
        for (int64_t e = 0; e < E; ++e) {                                                     \
            const T* xs = x + src[e] * dx;                                                    \
            const T* ye = y + e * dy;                                                         \
            T* od = out + dst[e] * dout;                                                      \
            for (int64_t j = 0; j < dout; ++j) {                                              \
                T v = APPLY_MOP(T, xs[xi[j]], ye[yi[j]], mop);                                \
                if (rop == R_SUM || rop == R_MEAN) od[j] += v;                                \
                else if (cnt[dst[e]] == 0) od[j] = v;                                         \
                else if (rop == R_MAX) od[j] = v > od[j] ? v : od[j];                         \
                else od[j] = v < od[j] ? v : od[j];                                           \
            }                                                                                 \
            if (cnt) cnt[dst[e]] += 1;                                                        \
        }                                                                                     \
        if (rop == R_MEAN) {                                                                  \
            for (int64_t r = 0; r < m; ++r) {                                                 \
                if (cnt[r] == 0) continue;                                                    \
                for (int64_t j = 0; j < dout; ++j)                                            \
                    out[r * dout + j] = out[r * dout + j] / (T)cnt[r];                        \
            }                                                                                 \
        }                                                                                     \
    }

DEF_SEND_UE_RECV(ref_send_ue_recv_f32, float)
DEF_SEND_UE_RECV(ref_send_ue_recv_f64, double)

/* send_uv (graph.py:964-966): out[e, j] = x[src[e], xi[j]] (mop) y[dst[e], yi[j]] */
#define DEF_SEND_UV(NAME, T)                                                                  \
    void NAME(const T* x, const T* y, const int64_t* src, const int64_t* dst, int64_t E,      \
              int64_t dx, int64_t dy, int64_t dout, const int32_t* xi, const int32_t* yi,     \
              int mop, T* out) {                                                              \
        for (int64_t e = 0; e < E; ++e) {                                                     \
            const T* xs = x + src[e] * dx;                                                    \
            const T* yd = y + dst[e] * dy;                                                    \
            for (int64_t j = 0; j < dout; ++j)                                                \
                out[e * dout + j] = APPLY_MOP(T, xs[xi[j]], yd[yi[j]], mop);                  \
        }                                                                                     \
    }

DEF_SEND_UV(ref_send_uv_f32, float)
DEF_SEND_UV(ref_send_uv_f64, double)

/* ------------------------------------------------------------------------------------------
 * segment_{sum,mean,max,min} (math.py:30-178 -> paddle.geometric.segment_*).
 *   data [E, d], ids [E] sorted non-decreasing int64; out [ids[E-1]+1, d] zero-filled first.
 * ---------------------------------------------------------------------------------------- */
#define DEF_SEGMENT(NAME, T)                                                                  \
    void NAME(const T* data, const int64_t* ids, int64_t E, int64_t d, int op, T* out) {      \
        if (E == 0) return;                                                                   \
        int64_t R = ids[E - 1] + 1;                                                           \
        memset(out, 0, sizeof(T) * (size_t)R * (size_t)d);                                    \
        int64_t s = 0;                                                                        \
        while (s < E) {                                                                       \
            int64_t t = s;                                                                    \
            while (t < E && ids[t] == ids[s]) ++t;                                            \
            T* o = out + ids[s] * d;                                                          \
            for (int64_t j = 0; j < d; ++j) o[j] = data[s * d + j];                           \
            for (int64_t p = s + 1; p < t; ++p) {                                             \
                const T* v = data + p * d;                                                    \
                if (op == R_SUM || op == R_MEAN) { for (int64_t j = 0; j < d; ++j) o[j] += v[j]; }            \
                else if (op == R_MAX) { for (int64_t j = 0; j < d; ++j) o[j] = v[j] > o[j] ? v[j] : o[j]; }   \
                else { for (int64_t j = 0; j < d; ++j) o[j] = v[j] < o[j] ? v[j] : o[j]; }                    \
            }                                                                                 \
            if (op == R_MEAN) for (int64_t j = 0; j < d; ++j) o[j] = o[j] / (T)(t - s);       \
            s = t;                                                                            \
        }                                                                                     \
    }

DEF_SEGMENT(ref_segment_f32, float)
DEF_SEGMENT(ref_segment_f64, double)
DEF_SEGMENT(ref_segment_i64, int64_t)

/* segment_softmax (math.py:216-224): m = segmax[id]; e = exp(x - m); e / segsum(e)[id] */
#define DEF_SEGMENT_SOFTMAX(NAME, T, EXPF)                                                    \
    void NAME(const T* data, const int64_t* ids, int64_t E, int64_t d, T* out) {              \
        int64_t s = 0;                                                                        \
        while (s < E) {                                                                       \
            int64_t t = s;                                                                    \
            while (t < E && ids[t] == ids[s]) ++t;                                            \
            for (int64_t j = 0; j < d; ++j) {                                                 \
                T mx = data[s * d + j];                                                       \
                for (int64_t p = s + 1; p < t; ++p)                                           \
                    mx = data[p * d + j] > mx ? data[p * d + j] : mx;                         \
                T sum = 0;                                                                    \
                for (int64_t p = s; p < t; ++p) {                                             \
                    T ev = EXPF(data[p * d + j] - mx);                                        \
                    out[p * d + j] = ev;                                                      \
                    sum += ev;                                                                \
                }                                                                             \
                for (int64_t p = s; p < t; ++p) out[p * d + j] = out[p * d + j] / sum;        \
            }                                                                                 \
            s = t;                                                                            \
        }                                                                                     \
    }

DEF_SEGMENT_SOFTMAX(ref_segment_softmax_f32, float, expf)
DEF_SEGMENT_SOFTMAX(ref_segment_softmax_f64, double, exp)

/* ------------------------------------------------------------------------------------------
 * build_index restatement (graph_kernel.pyx:59-88): stable counting sort by u.
 * Outputs int64, identical layout to the reference: degree[N], sorted_v[E], sorted_u[E],
 * sorted_eid[E], indptr[N+1].
 * ---------------------------------------------------------------------------------------- */
void ref_build_index(const int64_t* u, const int64_t* v, int64_t E, int64_t N, int64_t* degree,
                     int64_t* sorted_v, int64_t* sorted_u, int64_t* sorted_eid,
                     int64_t* indptr) {
    int64_t* count = (int64_t*)calloc((size_t)(N > 0 ? N : 1), sizeof(int64_t));
    memset(degree, 0, sizeof(int64_t) * (size_t)N);
    for (int64_t i = 0; i < E; ++i) degree[u[i]] += 1;
    indptr[0] = 0;
    for (int64_t i = 0; i < N; ++i) indptr[i + 1] = indptr[i] + degree[i];
    for (int64_t i = 0; i < E; ++i) {
        int64_t p = indptr[u[i]] + count[u[i]];
        sorted_v[p] = v[i];
        sorted_eid[p] = i;
        sorted_u[p] = u[i];
        count[u[i]] += 1;
    }
    free(count);
}

/* unique_segment on sorted keys (helper.py:156-160 -> paddle.unique(return_inverse=True)):
 * uniq[k] distinct values ascending, inv[e] = dense rank.  Returns number of distinct keys. */
int64_t ref_unique_segment(const int64_t* keys_sorted, int64_t E, int64_t* uniq, int64_t* inv) {
    int64_t k = -1;
    for (int64_t e = 0; e < E; ++e) {
        if (e == 0 || keys_sorted[e] != keys_sorted[e - 1]) uniq[++k] = keys_sorted[e];
        inv[e] = k;
    }
    return k + 1;
}
