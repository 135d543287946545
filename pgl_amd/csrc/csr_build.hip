// csr_build.hip -- K8: COO -> CSR on the GPU, bit-identical to the reference's build_index
// (pgl/graph_kernel.pyx:59-88: stable counting sort by key, ascending original edge id inside a
// row), plus unique_segment (pgl/utils/helper.py:156-160) and the seg_ptr helper.
//
// Plan (all HBM-bound integer work, ~28 B/edge algorithmic):
//   1. narrow the int64 keys (strided: a column of the [E,2] edge array) to int32, eid = iota;
//   2. stable LSD radix sort over ceil(log2 N) key bits only -- HAND-WRITTEN for this job (round 3; rounds 1-2 called
//      rocPRIM's radix_sort_pairs: three onesweep passes at 0.09 of the byte model): ceil(bits / 10) passes of <= 10 bits,
//      i.e. TWO passes up to 1 M rows, three up to 1 G, each pass = block histograms -> scan -> stable scatter (see "the sort" below);
//      the first pass reads the caller's int64 (u, v) columns directly and the last one writes the int32 (row, col, eid)
//      arrays the kernels read, so the narrow and unpack passes of the library version are gone;
//      stability == ascending eid inside equal keys == the reference's order by construction;
//   3. indptr from row boundaries in the sorted keys (no atomics, no scan): every position p with
//      key[p] != key[p-1] writes indptr for the rows in (key[p-1], key[p]];  degree = diff;
//   4. the neighbour id v and the original edge id ride THROUGH the sort as two more int32 streams next to the key (12 B per item
//      and pass, all sequential) instead of a random 8-byte gather v[eid] per edge afterwards (one 128-byte line per edge:
//      0.40 ms of the 1.04 at C2); the last pass writes the int32 (row, col, eid) arrays the aggregation kernels read and,
//      where asked, the reference's int64 arrays.
#include "common.hpp"
#include "scan.hpp"

#include <atomic>
#include <type_traits>

namespace pglamd {

static int key_bits(int64_t n) {
    int b = 1;
    while (b < 32 && (int64_t(1) << b) < n) ++b;
    return b;
}

__global__ __launch_bounds__(kBlock) void narrow_i64_kernel(const int64_t* __restrict__ in, int64_t stride, int64_t n, int32_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) out[i] = (int32_t)in[i * stride];
}

// indptr[r] = first position whose key >= r  (keys sorted); also covers r in (last key, n_rows]
// A position p that opens a new key writes indptr[prev + 1 .. cur] = p.  Short gaps (almost all of them) are written by the lane
// that found them; a LONG gap -- the rows after the last key of a sampled block, whose index spans the node space of the whole
// graph while only the batch's destinations have edges: tens of thousands of rows, 0.2 - 0.5 ms when one lane wrote them, 40 % of the
// GPU time of a mini-batch GraphSAGE step -- is written by the whole wave, 64 rows per store.
// Round 5: a lane takes kBoundsVec CONSECUTIVE positions (one 16-byte load of int32 keys; the key before them comes from the lane below)
// instead of one position and two 4-byte loads: the whole build 0.59 -> 0.56 ms at 20 M edges, 2.42 -> 2.30 ms at 100 M, csr_from_sorted
// 0.178 -> 0.155 / 0.84 -> 0.72 ms (profiles/r05/csr_build.txt against profiles/r04/csr_build.txt).
constexpr int kBoundsVec = 4;

template <typename K>
__global__ __launch_bounds__(kBlock) void row_bounds_kernel(const K* __restrict__ key, int64_t n, int64_t n_rows,
                                                            int64_t* __restrict__ indptr) {
    constexpr int64_t kLong = 16;
    constexpr int V = kBoundsVec;
    struct alignas(sizeof(K) * V) KV { K k[V]; };
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t step = (int64_t)gridDim.x * kBlock * V;
    const bool aligned = reinterpret_cast<uintptr_t>(key) % sizeof(KV) == 0;
    for (int64_t base = ((int64_t)blockIdx.x * kBlock + (threadIdx.x - lane)) * V; base <= n; base += step) {     // (wave-uniform trip count)
        const int64_t p0 = base + (int64_t)lane * V;              // this lane's positions p0 .. p0 + V - 1
        KV kv;
        if (aligned && p0 + V <= n) {
            kv = *reinterpret_cast<const KV*>(key + p0);
        } else {
#pragma unroll
            for (int i = 0; i < V; ++i) kv.k[i] = p0 + i < n ? key[p0 + i] : (K)0;
        }
        K before = (K)__shfl_up((long long)kv.k[V - 1], 1, kWave);           // the key at p0 - 1
        if (lane == 0) before = p0 > 0 && p0 - 1 < n ? key[p0 - 1] : (K)0;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int64_t p = p0 + i;
            int64_t lo = 0, hi = -1;                              // rows [lo, hi] get the value p
            if (p <= n) {
                lo = p == 0 ? 0 : (int64_t)(i == 0 ? before : kv.k[i - 1]) + 1;
                hi = p == n ? n_rows : (int64_t)kv.k[i];
                if (hi > n_rows) hi = n_rows;
            }
            const bool is_long = hi - lo >= kLong;
            if (!is_long)
                for (int64_t r = lo; r <= hi; ++r) indptr[r] = p;
            unsigned long long m = __ballot(is_long);
            while (m) {
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                const int64_t glo = __shfl(lo, l, kWave), ghi = __shfl(hi, l, kWave), gp = base + (int64_t)l * V + i;
                for (int64_t r = glo + lane; r <= ghi; r += kWave) indptr[r] = gp;
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void degree_kernel(const int64_t* __restrict__ indptr, int64_t n_rows,
                                                        int64_t* __restrict__ degree) {
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * kBlock)
        degree[r] = indptr[r + 1] - indptr[r];
}

struct NonEmpty {
    const int64_t* degree;
    __device__ int64_t operator()(int64_t r) const { return degree[r] > 0 ? 1 : 0; }
};

__global__ __launch_bounds__(kBlock) void uniq_scatter_kernel(const int64_t* __restrict__ degree, const int64_t* __restrict__ rank,
                                                              int64_t n_rows, int64_t* __restrict__ uniq, int64_t* __restrict__ num_uniq) {
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * kBlock) {
        if (degree[r] > 0) uniq[rank[r]] = r;
        if (r == n_rows - 1) *num_uniq = rank[r] + (degree[r] > 0 ? 1 : 0);
    }
}

__global__ __launch_bounds__(kBlock) void seg_ids_kernel(const int64_t* __restrict__ sorted_u, const int64_t* __restrict__ rank,
                                                         int64_t n, int64_t* __restrict__ seg) {
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock)
        seg[p] = rank[sorted_u[p]];
}


// ------------------------------------------------------------------------------------------------
// The sort.  One pass moves (key32, value64) pairs by one digit of BITS bits, stably:
//   hist      every 512-thread block counts the digits of its tile of 4 096 items (LDS atomics) -> hist[digit][block]
//             (measured, profiles/r03/csr_build_variants.txt: 512 x 8 items -- two to three blocks per CU -- 0.63 / 2.5 ms at C2 / C2';
//              1024 x 8: 0.64 / 3.1; 256 x 16: 0.76; tiles of 2 048 items: 0.95 - 1.0 ms, their runs per digit are too short)
//   scan      exclusive scan of hist in (digit, block) order: one block per digit scans its row, one small kernel scans the
//             row totals -> the global position of the first item of every (digit, block)
//   scatter   the block reloads its tile; wave w owns items [w * 512, (w + 1) * 512) of it and walks them 64 at a time IN
//             ORDER.  Inside a 64-item step the lanes holding the same digit find each other with BITS ballots (peers = AND
//             of ballot(bit) or its complement), rank = popcount of the peers below the lane, and the lowest peer bumps the
//             wave's own 16-bit digit counter in LDS by the group size -- no atomics, and the order inside a wave is the
//             item order.  A 16-step pass over the wave counters turns them into exclusive prefixes over the waves, so
//             position = hist offset + prefix over earlier waves + count in earlier steps of this wave + rank: stable.
//             The tile is then staged through LDS in block-sorted order and written out with consecutive lanes on
//             consecutive slots of a digit's run; consecutive tiles run on the same XCD, so the seams of the runs meet in one L2.
// Nothing here is graph specific except FIRST (keys / values come from the caller's strided int64 columns, ids are range
// checked) and LAST (the outputs are the CSR arrays).
// ------------------------------------------------------------------------------------------------
#ifndef PGLAMD_SORT_THREADS          // (variant builds: scripts/prof.py variant NAME PGLAMD_SORT_THREADS=512 ...)
#define PGLAMD_SORT_THREADS 512
#endif
#ifndef PGLAMD_SORT_ABLATE           // timing experiments of the scatter kernel (variant builds; the results are WRONG by construction):
#define PGLAMD_SORT_ABLATE 0          // 2 = no global stores, 4 = the loads alone -- profiles/r05/csr_scatter_ablation.txt
#endif
#ifndef PGLAMD_SORT_ITEMS
#define PGLAMD_SORT_ITEMS 8
#endif
constexpr int kSortThreads = PGLAMD_SORT_THREADS;
constexpr int kSortWaves = kSortThreads / kWave;
constexpr int kSortItems = PGLAMD_SORT_ITEMS;
constexpr int kSortTile = kSortThreads * kSortItems;
constexpr int kSortMaxBits = 10;

struct SortArgs {
    const int64_t* u; int64_t us; const int64_t* v; int64_t vs;      // FIRST: strided int64 key / neighbour columns
    const int32_t* key_in; const int32_t* col_in; const int32_t* eid_in;     // otherwise: three int32 streams (key, neighbour, original edge id)
    int32_t* key_out; int32_t* col_out; int32_t* eid_out;                   // not LAST
    int32_t* row32; int32_t* col32; int32_t* eid32;                  // LAST (any may be NULL)
    int64_t* sorted_u; int64_t* sorted_v; int64_t* sorted_eid;       // LAST, optional int64 copies (the reference's arrays)
    uint32_t* hist;                                                  // [bins][nblk]
    uint32_t* dbase;                                                 // [bins]
    int32_t* range_flag;
    int64_t n, n_rows;
    int shift, nblk;
    uint32_t* status;      // one-sweep: [tile][bins] look-back words, {flag:2, count:30}
    uint32_t* ticket;      // one-sweep: the pass's tile counter
    int group;             // one-sweep: tiles per XCD group (tile_of_ticket)
    int clear_status;      // one-sweep: the launcher clears this pass's look-back words first (big builds; small ones clear all passes at once)
    int pair;              // FIRST: 0 = separate strided columns; 1 / 2 = u and v are the two columns of ONE [E, 2] int64 array
                           // (u first / v first): both come from a single 16-byte load per edge
};

struct alignas(16) I64x2 { int64_t a, b; };

template <bool FIRST>
__device__ __forceinline__ int32_t sort_key(const SortArgs& a, int64_t idx, bool& bad) {
    if constexpr (FIRST) {
        int64_t k = a.u[idx * a.us];
        if ((uint64_t)k >= (uint64_t)a.n_rows) { k = 0; bad = true; }   // clamped (memory-safe) and reported
        return (int32_t)k;
    } else {
        return a.key_in[idx];
    }
}

// logical tile of a block: consecutive tiles run on the SAME XCD (dispatch is round-robin over the 8 XCDs), so that the
// partial cache lines two neighbouring tiles write at the seam of a digit's run meet in one L2
__device__ __forceinline__ int64_t sort_tile(int nblk) { return xcd_swizzle(blockIdx.x, nblk); }

// Round 5: persistent blocks.  One block per tile lived ~10 us, most of it the latency of its one round of loads and of its write-out with
// nothing else in flight (80 MB of int32 keys took 47 us = 1.7 TB/s at 20 M edges).  Now kHistBlocksPerCu blocks per CU walk the tiles of
// their XCD's share (consecutive tiles stay on one XCD: hist[d][t] and hist[d][t + 1] are neighbours in memory), the keys of the NEXT
// tile already loading while the current one is counted and written out.  Measured: int32 keys 47 -> 40 us at 20 M edges, 162 -> 147 us at
// 100 M; the first pass (int64 keys at a 16-byte stride: 5.3 TB/s of line traffic at 100 M edges) does not move.
constexpr int kHistBlocksPerCu = 4;

template <int BITS, bool FIRST>
__global__ __launch_bounds__(kSortThreads) void sort_hist_kernel(SortArgs a) {
    constexpr int BINS = 1 << BITS;
    __shared__ uint32_t h[BINS];
    // XCD x = blockIdx % 8 owns tiles [x * per, (x + 1) * per); its blocks take them round-robin
    const int64_t per = (a.nblk + kXcds - 1) / kXcds;
    const int64_t xcd = blockIdx.x % kXcds, nbx = (gridDim.x - xcd + kXcds - 1) / kXcds;
    const int64_t lo = xcd * per, hi = lo + per < a.nblk ? lo + per : a.nblk;
    for (int i = threadIdx.x; i < BINS; i += kSortThreads) h[i] = 0;
    bool bad = false;
    int32_t cur[kSortItems], nxt[kSortItems];
    auto load = [&](int64_t tile, int32_t (&k)[kSortItems]) {
        const int64_t base = tile * kSortTile;
#pragma unroll
        for (int i = 0; i < kSortItems; ++i) {
            const int64_t idx = base + (int64_t)i * kSortThreads + threadIdx.x;
            k[i] = tile < hi && idx < a.n ? sort_key<FIRST>(a, idx, bad) : -1;
        }
    };
    int64_t tile = lo + blockIdx.x / kXcds;
    load(tile, cur);
    __syncthreads();
    for (; tile < hi; tile += nbx) {
        load(tile + nbx, nxt);
#pragma unroll
        for (int i = 0; i < kSortItems; ++i)
            if (cur[i] >= 0) atomicAdd(&h[(cur[i] >> a.shift) & (BINS - 1)], 1u);
        __syncthreads();
        for (int i = threadIdx.x; i < BINS; i += kSortThreads) { a.hist[(int64_t)i * a.nblk + tile] = h[i]; h[i] = 0; }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kSortItems; ++i) cur[i] = nxt[i];
    }
    if (FIRST && a.range_flag && __any(bad)) { if ((threadIdx.x & (kWave - 1)) == 0) atomicOr(a.range_flag, 1); }
}

// Exclusive scan of hist in (digit, tile) order = the flattened [digit][tile] array: the result IS the global position of the
// first item of every (digit, tile).  Three small coalesced kernels (piece sums -> scan of the piece sums -> scan inside the
// pieces); the first version scanned one digit's row per block with every thread on its own 19-entry segment -- uncoalesced, 48 us
// per pass at 20 M edges, 15 % of the whole build.
constexpr int kScanPiece = 2048;          // entries per block (256 threads x 8)

__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* wave_tot, uint32_t& total) {
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) { const uint32_t t = __shfl_up(inc, off, kWave); if (lane >= off) inc += t; }
    __syncthreads();                                    // (wave_tot may still be read by the previous call)
    if (lane == kWave - 1) wave_tot[w] = inc;
    __syncthreads();
    uint32_t before = 0, tot = 0;
#pragma unroll
    for (int ww = 0; ww < kBlock / kWave; ++ww) { const uint32_t t = wave_tot[ww]; if (ww < w) before += t; tot += t; }
    total = tot;
    return before + inc - v;
}

__global__ __launch_bounds__(kBlock) void scan_piece_sums_kernel(const uint32_t* __restrict__ h, int64_t n, uint32_t* __restrict__ sums) {
    __shared__ uint32_t wave_tot[kBlock / kWave];
    const int64_t base = (int64_t)blockIdx.x * kScanPiece;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < kScanPiece / kBlock; ++i) { const int64_t j = base + i * kBlock + threadIdx.x; if (j < n) s += h[j]; }
    uint32_t total;
    (void)block_exclusive_scan_256(s, wave_tot, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(kBlock) void scan_sums_kernel(uint32_t* __restrict__ sums, int n) {       // one block: n <= a few 10^5
    __shared__ uint32_t wave_tot[kBlock / kWave];
    uint32_t carry = 0;
    for (int b = 0; b < n; b += kBlock) {
        const int j = b + threadIdx.x;
        const uint32_t v = j < n ? sums[j] : 0;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan_256(v, wave_tot, total);
        if (j < n) sums[j] = carry + ex;
        carry += total;
    }
}

__global__ __launch_bounds__(kBlock) void scan_apply_kernel(uint32_t* __restrict__ h, int64_t n, const uint32_t* __restrict__ sums) {
    __shared__ uint32_t wave_tot[kBlock / kWave];
    const int64_t base = (int64_t)blockIdx.x * kScanPiece;
    uint32_t carry = sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < kScanPiece / kBlock; ++i) {
        const int64_t j = base + i * kBlock + threadIdx.x;
        const uint32_t v = j < n ? h[j] : 0;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan_256(v, wave_tot, total);
        if (j < n) h[j] = carry + ex;
        carry += total;
    }
}

// The scatter stages the tile through LDS in block-sorted order, so that consecutive lanes write consecutive slots of a
// digit's run (a first version let every lane store its own triple where it belongs: correct, and 35 % SLOWER than the library
// sort it replaced -- 64 partial cache lines per store instruction).
// Round 4: the three 32-bit streams of an item (key, neighbour, edge id) go through ONE 16 KB stage one after the other instead
// of through a 48 KB (key32 + value64) stage at once.  Counters had shown the kernel moving exactly its model bytes (FETCH 345 MB,
// WRITE 262 MB at C2: the partial lines of neighbouring tiles do merge in the XCD's L2) at 2.8 TB/s -- it was neither byte- nor
// request-bound but occupancy-bound: 57 KB of LDS = two blocks per CU, whose load / rank / scan / stage / store phases are separated
// by barriers, so the memory pipes idled whenever both blocks were ranking (profiles/r04/pmc_csr_build.txt).  24 KB = four blocks.
// Tried and removed in round 4: ranking with ONE returning LDS atomic per item on the wave's private counter instead of BITS ballots
// (the ballot ranking is 900 of this kernel's 1 300 vector-ALU instructions per wave).  It was bit-exact and stable on every test
// (the hardware serves the lanes of one ds_add_rtn in lane order; an on-device check of ascending edge ids inside every row guarded
// it) and bought nothing: 0.576 -> 0.561 ms at C2, 2.39 -> 2.49 ms at C2' (profiles/r04/csr_build_variants.txt) -- the kernel is not
// bound by its vector instructions either.
//
// SWEEP (round 5): the one-sweep form of a pass.  No per-tile histogram kernel and no scan kernels: a pass is this kernel alone, after ONE
// histogram of all the digits of all the passes (sweep_hist_kernel) whose exclusive scan over the digits (sweep_base_kernel) is a.dbase.
// A block draws its tile from a ticket counter (so every tile before it is running or done), counts its digits while ranking as before,
// publishes the counts in status[tile][digit] (flag 1 = this tile's count, flag 2 = count of this and ALL earlier tiles; the flag and the
// 30-bit count share one 32-bit word, so a relaxed agent-scope store / load is the whole protocol) and every thread walks back over the
// earlier tiles' words of ITS digit, adding counts until it meets a flag 2 (decoupled look-back), then publishes its own flag 2.
// position = dbase[digit] + count in earlier tiles + rank inside the tile: the same position the scanned per-tile histogram gives -- the
// output is bit-identical (tests/test_gpu_round5.py::test_csr_onesweep_*).
constexpr uint32_t kSweepCountMask = (1u << 30) - 1;
#ifndef PGLAMD_SWEEP_BATCH
#define PGLAMD_SWEEP_BATCH 4
#endif
#ifndef PGLAMD_SWEEP_AUTO_MAX_EDGES
#define PGLAMD_SWEEP_AUTO_MAX_EDGES 1000000
#endif
constexpr int kSweepOneFillTiles = 1024;        // up to this many tiles the look-back words of all passes are cleared by ONE fill with the digit totals
constexpr int kSweepBatch = PGLAMD_SWEEP_BATCH;

// ticket -> tile.  Tickets are drawn in dispatch order and dispatch goes round the 8 XCDs, so ticket t runs on XCD t % 8 (a performance
// assumption only).  `group` consecutive tiles go to one XCD: the short runs neighbouring tiles write to the same digit (4 items = 16 B at
// 10-bit digits) meet in that XCD's L2 and leave it as whole lines.  Inside a window of 8 * group tickets a tile may wait for a tile whose
// ticket is up to 8 * group later, so the host keeps 8 * group far below the number of resident blocks (sweep_group); group 1 is plain
// ticket order.  The tail that does not fill a window is taken in ticket order.
__device__ __forceinline__ int64_t tile_of_ticket(uint32_t t, int nblk, int group) {
    const uint32_t win = 8u * (uint32_t)group;
    if (group <= 1 || t >= ((uint32_t)nblk / win) * win) return t;
    const uint32_t w = t / win, r = t % win;
    return (int64_t)w * win + (r % 8u) * (uint32_t)group + r / 8u;
}

// Round 5, what bounds this kernel (profiles/r05/csr_scatter_ablation.txt; pass 0 / pass 1 at 20 M edges): the tile's loads alone 73 / 54 us
// (4.4 TB/s: sequential reads), loads + ranking + LDS staging 120 / 91 us, everything 208 / 156 us -- the store phase adds 88 / 65 us for 240 MB =
// 2.7 TB/s.  A PERSISTENT form (two blocks per CU walking their XCD's tiles, the next tile's 24 loads + histogram row prefetched into a second
// register set before the current tile is ranked; bit-exact) was built to overlap the three: 221 / 159 us, and 867 against 640 us on the first
// pass at 100 M edges -- slower.  Its loads-only time is the same 79 us and its stores still add 65 us: reads and writes share the HBM, and
// 3 072 concurrent write streams (1 024 digit runs x 3 arrays, 16-byte pieces merged to 64-byte requests in L2) drain at 2.7 TB/s whatever
// else is in flight.  A second persistent form (reads split around the ranking: neighbour / edge ids before it, the next tile's keys after it)
// took 0.645 / 2.89 ms for the whole build against 0.553 / 2.30.  The sort is bound by what the memory system gives this mix of sequential
// reads and scattered writes, not by the phases between a tile's loads and stores; it stays one tile per block
// (csrc/variants/csr_build_persistent_scatter*.patch).
template <int BITS, bool FIRST, bool LAST, bool SWEEP>
__global__ __launch_bounds__(kSortThreads) void sort_scatter_kernel(SortArgs a) {
    constexpr int BINS = 1 << BITS;
    constexpr int PER = BINS / kSortThreads > 0 ? BINS / kSortThreads : 1;     // digits per thread in the block-level scans
    using CntT = uint16_t;
    constexpr size_t kCntBytes = sizeof(CntT) * kSortWaves * BINS, kStageBytes = sizeof(int32_t) * kSortTile;
    __shared__ uint32_t gb[BINS];                         // global position of this block's first item of every digit
    __shared__ uint32_t dstart[BINS];                     // position of every digit's first item in the block-sorted tile
    __shared__ __align__(16) unsigned char lds_pool[kCntBytes > kStageBytes ? kCntBytes : kStageBytes];
    __shared__ uint32_t wave_tot[kSortWaves];
    int32_t* stage = reinterpret_cast<int32_t*>(lds_pool);                     // one 32-bit stream of the tile at a time ...
    CntT (*cnt)[BINS] = reinterpret_cast<CntT (*)[BINS]>(lds_pool);           // ... over the per-wave digit counters: [kSortWaves][BINS] counts, then prefixes over waves
    __shared__ int64_t s_tile;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), w = tid >> 6;
    if constexpr (SWEEP) {
        if (tid == 0) s_tile = tile_of_ticket(atomicAdd(a.ticket, 1u), a.nblk, a.group);
    } else {
        if (tid == 0) s_tile = sort_tile(a.nblk);
    }
    {
        uint32_t* z = reinterpret_cast<uint32_t*>(&cnt[0][0]);
        for (int i = tid; i < (int)(kCntBytes / 4); i += kSortThreads) z[i] = 0;
    }
    __syncthreads();
    const int64_t tile = s_tile;
    if (tile < 0) return;                                  // (block-uniform)
    for (int i = tid; i < BINS; i += kSortThreads) gb[i] = SWEEP ? 0u : a.hist[(int64_t)i * a.nblk + tile];       // (already the global position: flat scan)
    if constexpr (SWEEP) __syncthreads();
    const int64_t tbase = tile * kSortTile;
    const int64_t wbase = tbase + (int64_t)w * (kWave * kSortItems);
    int32_t key[kSortItems], col[kSortItems], eid[kSortItems];
    int pos[kSortItems];
    bool bad = false;
#pragma unroll
    for (int s = 0; s < kSortItems; ++s) {
        const int64_t idx = wbase + s * kWave + lane;
        key[s] = 0; col[s] = 0; eid[s] = 0;
        if (idx < a.n) {
            if constexpr (FIRST) {
                int64_t k, nb;
                if (a.pair) {                       // one 16-byte load for both columns of the edge array
                    const I64x2 e = reinterpret_cast<const I64x2*>(a.pair == 1 ? a.u : a.v)[idx];
                    k = a.pair == 1 ? e.a : e.b; nb = a.pair == 1 ? e.b : e.a;
                } else {
                    k = a.u[idx * a.us]; nb = a.v[idx * a.vs];
                }
                if ((uint64_t)k >= (uint64_t)a.n_rows) { k = 0; bad = true; }              // clamped (memory-safe) and reported
                if ((uint64_t)nb > (uint64_t)INT32_MAX) bad = true;
                key[s] = (int32_t)k; col[s] = (int32_t)nb; eid[s] = (int32_t)idx;         // (neighbour, original edge id)
            } else {
                key[s] = a.key_in[idx]; col[s] = a.col_in[idx]; eid[s] = a.eid_in[idx];
            }
        }
    }
    if (FIRST && a.range_flag && __any(bad)) { if (lane == 0) atomicOr(a.range_flag, 1); }
#if PGLAMD_SORT_ABLATE == 4      // (timing experiment, wrong results: the tile's loads alone)
    { int acc = 0;
#pragma unroll
      for (int s = 0; s < kSortItems; ++s) acc ^= key[s] ^ col[s] ^ eid[s];
      if (acc != 0x7ffffff1) return; }
#endif
    if constexpr (SWEEP) {
        // The tile's digit counts are published BEFORE the ranking (one LDS atomic per item into gb, which is free until the look-back):
        // ranking is half of a block's life, and a tile cannot finish its look-back before every earlier tile has published its counts --
        // published after the ranking (version 1) the passes ran in convoys behind the slowest ranker, 1.7 x the multi-kernel scatter.
#pragma unroll
        for (int s = 0; s < kSortItems; ++s)
            if (wbase + s * kWave + lane < a.n) atomicAdd(&gb[(key[s] >> a.shift) & (BINS - 1)], 1u);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int d = tid * PER + j;
            if (d < BINS) __hip_atomic_store(a.status + tile * BINS + d, ((tile == 0 ? 2u : 1u) << 30) | gb[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // rank inside the wave, in item order: the lanes of a 64-item step that hold the same digit find each other with BITS
    // ballots; the lowest of them bumps the wave's counter of that digit by the group size
#pragma unroll
    for (int s = 0; s < kSortItems; ++s) {
        const bool valid = wbase + s * kWave + lane < a.n;
        const int d = (key[s] >> a.shift) & (BINS - 1);
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < BITS; ++b) {
            const bool bit = (d >> b) & 1;
            const unsigned long long m = __ballot(valid && bit);
            peers &= bit ? m : ~m;
        }
        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(peers >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)peers, 0));
        const int leader = __builtin_ctzll(peers | (1ull << 63));
        int old = 0;
        if (valid && rank == 0) { old = cnt[w][d]; cnt[w][d] = (uint16_t)(old + __popcll(peers)); }
        old = __shfl(old, leader, kWave);
        pos[s] = old + rank;
    }
    __syncthreads();
    // counts -> exclusive prefixes over the waves (per digit); block totals -> exclusive scan over the digits
    uint32_t tot[PER];
    {
        uint32_t sum = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int d = tid * PER + j;
            uint32_t run = 0;
            if (d < BINS) {
#pragma unroll
                for (int ww = 0; ww < kSortWaves; ++ww) { const uint32_t c = cnt[ww][d]; cnt[ww][d] = (CntT)run; run += c; }
            }
            tot[j] = run; sum += run;
        }
        uint32_t inc = sum;                               // inclusive scan of `sum` over the block: wave scan + wave totals
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) { const uint32_t t = __shfl_up(inc, off, kWave); if (lane >= off) inc += t; }
        if (lane == kWave - 1) wave_tot[w] = inc;
        __syncthreads();
        uint32_t before = 0;
        for (int ww = 0; ww < w; ++ww) before += wave_tot[ww];
        uint32_t run = before + inc - sum;
#pragma unroll
        for (int j = 0; j < PER; ++j) { const int d = tid * PER + j; if (d < BINS) dstart[d] = run; run += tot[j]; }
    }
    __syncthreads();
    // block-sorted position of every item (registers), then the stage may overwrite the counters
#pragma unroll
    for (int s = 0; s < kSortItems; ++s) {
        const int d = (key[s] >> a.shift) & (BINS - 1);
        pos[s] += (int)dstart[d] + (int)cnt[w][d];
    }
    __syncthreads();
    const int n_tile = (int)min((int64_t)kSortTile, a.n - tbase);
    // round 1: the keys.  Slot i of the block-sorted tile goes to dest[i]; the destinations stay in registers for rounds 2 and 3.
#pragma unroll
    for (int s = 0; s < kSortItems; ++s)
        if (wbase + s * kWave + lane < a.n) stage[pos[s]] = key[s];
    if constexpr (SWEEP) {
        // decoupled look-back, one digit per thread (PER digits at 10 bits): four earlier tiles' words per round trip
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int d = tid * PER + j;
            if (d >= BINS) continue;
            uint32_t before = 0;
            if (tile > 0) {
                int64_t k = tile - 1;
                bool done = false;
                uint32_t spins = 0;
                while (!done) {
                    uint32_t sw[kSweepBatch];
#pragma unroll
                    for (int i = 0; i < kSweepBatch; ++i)
                        sw[i] = k - i >= 0 ? __hip_atomic_load(a.status + (k - i) * BINS + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2u << 30);
                    int adv = 0;
#pragma unroll
                    for (int i = 0; i < kSweepBatch; ++i) {
                        if (done || adv != i) continue;                  // (stopped at a word that is not published yet: re-read from there)
                        const uint32_t f = sw[i] >> 30;
                        if (f == 0) continue;
                        before += sw[i] & kSweepCountMask;
                        ++adv;
                        if (f == 2) done = true;
                    }
                    k -= adv;
                    if (!done && adv == 0) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > (1u << 26)) __builtin_trap();      // (seconds: a predecessor that never ran -- fail loudly, never hang the device)
                    }
                }
                __hip_atomic_store(a.status + tile * BINS + d, (2u << 30) | (before + tot[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            gb[d] = a.dbase[d] + before;
        }
    }
    __syncthreads();
    int32_t dest[kSortItems];
#pragma unroll
    for (int s = 0; s < kSortItems; ++s) {
        const int i = s * kSortThreads + tid;
        dest[s] = -1;
        if (i >= n_tile) continue;
        const int32_t k = stage[i];
        const int d = (k >> a.shift) & (BINS - 1);
        dest[s] = (int32_t)(gb[d] + (uint32_t)(i - (int)dstart[d]));
#if PGLAMD_SORT_ABLATE == 2      // (timing experiment, wrong results: everything but the global stores)
        if (dest[s] != 0x7ffffff0) { dest[s] = -1; continue; }
#endif
        if constexpr (LAST) {
            if (a.row32) a.row32[dest[s]] = k;
            if (a.sorted_u) a.sorted_u[dest[s]] = k;
        } else {
            a.key_out[dest[s]] = k;
        }
    }
    __syncthreads();
    // round 2: the neighbour ids
#pragma unroll
    for (int s = 0; s < kSortItems; ++s)
        if (wbase + s * kWave + lane < a.n) stage[pos[s]] = col[s];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kSortItems; ++s) {
        if (dest[s] < 0) continue;
        const int32_t c = stage[s * kSortThreads + tid];
        if constexpr (LAST) {
            if (a.col32) a.col32[dest[s]] = c;
            if (a.sorted_v) a.sorted_v[dest[s]] = c;
        } else {
            a.col_out[dest[s]] = c;
        }
    }
    __syncthreads();
    // round 3: the original edge ids
#pragma unroll
    for (int s = 0; s < kSortItems; ++s)
        if (wbase + s * kWave + lane < a.n) stage[pos[s]] = eid[s];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kSortItems; ++s) {
        if (dest[s] < 0) continue;
        const int32_t e = stage[s * kSortThreads + tid];
        if constexpr (LAST) {
            if (a.eid32) a.eid32[dest[s]] = e;
            if (a.sorted_eid) a.sorted_eid[dest[s]] = e;
        } else {
            a.eid_out[dest[s]] = e;
        }
    }
}

// One-sweep, step 1: the histogram of every pass's digit over ALL keys, in one read of the key column.  At most 1 024 blocks, each over a
// strided set of tiles with all the passes' bins in LDS (4 passes x 10 bits = 16 KB at most), flushed with one global atomic per non-empty bin.
struct SweepHistArgs {
    const int64_t* u; int64_t us;
    int64_t n, n_rows;
    int nblk, passes;
    int shift[4], width[4];
    uint32_t* ghist;               // [sum of 2^width]: pass p starts at the sum of the earlier passes' bins
};

__global__ __launch_bounds__(kSortThreads) void sweep_hist_kernel(SweepHistArgs a) {
    __shared__ uint32_t h[4 << kSortMaxBits];
    int off[4], total = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) { off[p] = total; if (p < a.passes) total += 1 << a.width[p]; }
    for (int i = threadIdx.x; i < total; i += kSortThreads) h[i] = 0;
    __syncthreads();
    for (int64_t tile = blockIdx.x; tile < a.nblk; tile += gridDim.x) {
        const int64_t base = tile * kSortTile;
        int64_t k[kSortItems];
#pragma unroll
        for (int i = 0; i < kSortItems; ++i) {
            const int64_t idx = base + (int64_t)i * kSortThreads + threadIdx.x;
            k[i] = idx < a.n ? a.u[idx * a.us] : -1;
        }
#pragma unroll
        for (int i = 0; i < kSortItems; ++i) {
            if (base + (int64_t)i * kSortThreads + threadIdx.x >= a.n) continue;
            const uint32_t key = (uint64_t)k[i] >= (uint64_t)a.n_rows ? 0u : (uint32_t)k[i];       // (clamped exactly as the pass clamps it)
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (p < a.passes) atomicAdd(&h[off[p] + ((key >> a.shift[p]) & ((1u << a.width[p]) - 1))], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < total; i += kSortThreads) { const uint32_t c = h[i]; if (c) atomicAdd(a.ghist + i, c); }
}

// step 2: block p turns pass p's digit totals into the position of every digit's first item (exclusive scan over the digits)
__global__ __launch_bounds__(kBlock) void sweep_base_kernel(SweepHistArgs a, uint32_t* __restrict__ dbase) {
    __shared__ uint32_t wave_tot[kBlock / kWave];
    int off = 0;
    for (int p = 0; p < (int)blockIdx.x; ++p) off += 1 << a.width[p];
    const int bins = 1 << a.width[blockIdx.x];
    uint32_t carry = 0;
    for (int b = 0; b < bins; b += kBlock) {
        const int j = b + threadIdx.x;
        const uint32_t v = j < bins ? a.ghist[off + j] : 0;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan_256(v, wave_tot, total);
        if (j < bins) dbase[off + j] = carry + ex;
        carry += total;
    }
}

// One-sweep or multi-kernel passes?  Returns the tiles per XCD group of the one-sweep passes (tile_of_ticket), 0 = multi-kernel passes.
// Option "csr_onesweep" (pglamd_set_option; first value from PGLAMD_CSR_ONESWEEP): -1 (default) by size, 0 never, g >= 1 always, group g
// -- capped so that a window of 8 * group tickets stays below a sixteenth of the blocks the device holds at two per CU (a tile may wait for
// a ticket up to one window later).
// By size = up to 1 M edges (profiles/r05/csr_onesweep.txt): the one-sweep build is 7 launches instead of 12 and wins where launches
// dominate -- 0.038 against 0.054 ms at Cora's size, 0.041 / 0.060 at 100 k edges, 0.060 / 0.071 at a sampled block of 600 k edges, level
// at 1 M -- and loses on big graphs: 0.095 / 0.086 ms at 2 M edges, 0.74 / 0.60 at 20 M, 2.8 / 2.4 at 100 M.  Counters say why: tiles in ticket order go round the XCDs, so the 16-byte runs neighbouring tiles write to one
// digit no longer meet in one L2 -- 12.2 M write requests, half of them 32-byte, 570 MB written, where the multi-kernel scatter (an XCD
// takes consecutive tiles) issues 4.2 M, 94 % full 64-byte, 256 MB.  Giving an XCD groups of consecutive tiles (group 4 .. 32) restores
// the merging but makes the first tile of a group wait for tickets drawn later: slower still (0.78 - 0.82 ms).  Publishing the counts
// before the ranking, wider look-back reads (16 words per round trip) and a path-compressed look-back did not change the picture
// (profiles/r05/csr_onesweep_v1..v3.txt).
constexpr int64_t kSweepAutoMaxEdges = PGLAMD_SWEEP_AUTO_MAX_EDGES;

static int sweep_group(int64_t E) {
    static int cap = [] {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return 1; }
        const int c = prop.multiProcessorCount / 8;
        return c < 1 ? 1 : c;
    }();
    if (E >= ((int64_t)1 << 30)) return 0;                 // (the look-back words carry 30-bit counts)
    int g = csr_onesweep_option().load(std::memory_order_relaxed);
    if (g < 0) return E <= kSweepAutoMaxEdges ? 1 : 0;
    return g > cap ? cap : g;
}

// Widest digit a pass may take.  Measured (profiles/r03/csr_build.txt): 20-bit keys sort faster in 2 passes of 10 bits (0.64 ms at
// 20 M edges) than in 3 of 7 (0.69); 22-bit keys faster in 3 passes of 8 / 7 / 7 (3.1 ms at 100 M edges) than in 2 of 11 (4.4):
// at 11 bits a tile of 8 192 items leaves runs of four items per digit, too short to write whole cache lines.
static int sort_max_bits() {
    static int v = [] { const char* e = getenv("PGLAMD_SORT_MAXBITS"); int b = e ? atoi(e) : 10; return b < 6 ? 6 : b > kSortMaxBits ? kSortMaxBits : b; }();
    return v;
}

static int sort_cus() {
    static int v = [] {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
        return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }();
    return v;
}

template <int BITS, bool FIRST, bool LAST>
static int32_t sort_pass_launch(const SortArgs& a, uint32_t* totals, hipStream_t st) {
    const unsigned grid = (unsigned)xcd_grid(a.nblk);
    const unsigned hist_grid = grid < (unsigned)(sort_cus() * kHistBlocksPerCu) ? grid : (unsigned)(sort_cus() * kHistBlocksPerCu);
    hipLaunchKernelGGL((sort_hist_kernel<BITS, FIRST>), dim3(hist_grid), dim3(kSortThreads), 0, st, a);
    PGLAMD_LAUNCH_CHECK();
    const int64_t n_hist = (int64_t)(1 << BITS) * a.nblk;
    const int pieces = (int)ceil_div(n_hist, (int64_t)kScanPiece);
    hipLaunchKernelGGL(scan_piece_sums_kernel, dim3((unsigned)pieces), dim3(kBlock), 0, st, a.hist, n_hist, totals);
    PGLAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kBlock), 0, st, totals, pieces);
    PGLAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)pieces), dim3(kBlock), 0, st, a.hist, n_hist, totals);
    PGLAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL((sort_scatter_kernel<BITS, FIRST, LAST, false>), dim3(grid), dim3(kSortThreads), 0, st, a);
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}

template <int BITS, bool FIRST, bool LAST>
static int32_t sweep_pass_launch(const SortArgs& a, hipStream_t st) {
    if (a.clear_status) PGLAMD_HIP_CHECK(hipMemsetAsync(a.status, 0, (size_t)a.nblk * sizeof(uint32_t) << BITS, st));
    hipLaunchKernelGGL((sort_scatter_kernel<BITS, FIRST, LAST, true>), dim3((unsigned)a.nblk), dim3(kSortThreads), 0, st, a);
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}

template <bool FIRST, bool LAST>
static int32_t sweep_pass(int bits, const SortArgs& a, hipStream_t st) {
    switch (bits) {
        case 6: return sweep_pass_launch<6, FIRST, LAST>(a, st);
        case 7: return sweep_pass_launch<7, FIRST, LAST>(a, st);
        case 8: return sweep_pass_launch<8, FIRST, LAST>(a, st);
        case 9: return sweep_pass_launch<9, FIRST, LAST>(a, st);
        default: return sweep_pass_launch<10, FIRST, LAST>(a, st);
    }
}

template <bool FIRST, bool LAST>
static int32_t sort_pass(int bits, const SortArgs& a, uint32_t* totals, hipStream_t st) {
    switch (bits) {
        case 6: return sort_pass_launch<6, FIRST, LAST>(a, totals, st);
        case 7: return sort_pass_launch<7, FIRST, LAST>(a, totals, st);
        case 8: return sort_pass_launch<8, FIRST, LAST>(a, totals, st);
        case 9: return sort_pass_launch<9, FIRST, LAST>(a, totals, st);
        default: return sort_pass_launch<10, FIRST, LAST>(a, totals, st);
    }
}

// digits of the passes for keys of `bits` bits: as few passes as the widest digit allows, digits of (nearly) equal width >= 6
static int sort_plan(int bits, int (&width)[8]) {
    const int mb = sort_max_bits();
    int passes = (bits + mb - 1) / mb;
    if (passes < 1) passes = 1;
    int left = bits;
    for (int p = 0; p < passes; ++p) {
        int w = (left + (passes - p) - 1) / (passes - p);
        if (w < 6) w = 6;
        width[p] = w;
        left -= w;
        if (left < 0) left = 0;
    }
    return passes;
}

// small builds (up to kSweepOneFillTiles tiles): the look-back words of ALL passes, cleared by one fill together with the digit totals
static size_t sweep_small_words(int64_t E) {
    const size_t n_tiles = (size_t)ceil_div(E > 0 ? E : 1, (int64_t)kSortTile);
    return n_tiles <= (size_t)kSweepOneFillTiles ? n_tiles * ((size_t)4 << kSortMaxBits) : 0;
}

static size_t sort_hist_entries(int64_t E) { return ((size_t)1 << kSortMaxBits) * (size_t)ceil_div(E > 0 ? E : 1, (int64_t)kSortTile); }

static unsigned grid_for(int64_t n) {
    int64_t g = ceil_div(n > 0 ? n : 1, kBlock);
    return (unsigned)(g < 256 * 16 ? g : 256 * 16);
}

static size_t scan_temp_bytes(int64_t N) { return exclusive_scan64_temp_bytes(N); }

}  // namespace pglamd

using namespace pglamd;

extern "C" size_t pglamd_csr_build_workspace_bytes(int64_t num_edges, int64_t num_nodes) {
    const int64_t E = num_edges > 0 ? num_edges : 1;
    // two (key, neighbour, edge id) int32 ping-pong sets, row32 when the caller does not keep it, block histograms + digit totals / bases
    (void)num_nodes;
    return 7 * align_up((size_t)E * 4, 256) + align_up(sort_hist_entries(E) * 4, 256) +
           align_up((size_t)(ceil_div((int64_t)sort_hist_entries(E), (int64_t)kScanPiece) + 1) * 4, 256) + align_up(((size_t)1 << kSortMaxBits) * 4, 256) + 1024 +
           2 * align_up(((size_t)4 << kSortMaxBits) * 4 + 64, 256) +               // one-sweep: digit totals of four passes + tickets, digit bases,
           align_up(sweep_small_words(E) * 4, 256);                                 //            look-back words of all passes of a small build
}

extern "C" int32_t pglamd_csr_build(const int64_t* u, int64_t u_stride, const int64_t* v, int64_t v_stride,
                                    int64_t num_edges, int64_t num_nodes, int64_t* degree, int64_t* sorted_v,
                                    int64_t* sorted_u, int64_t* sorted_eid, int64_t* indptr, int32_t* row32,
                                    int32_t* col32, int32_t* eid32, int32_t* range_flag, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    if (num_edges < 0 || num_nodes < 0 || num_edges >= INT32_MAX || num_nodes >= INT32_MAX)
        return fail(PGLAMD_E_RANGE, "csr_build: E=%lld N=%lld beyond the int32 engine range", (long long)num_edges, (long long)num_nodes);
    if (!indptr || (num_nodes > 0 && !degree) || (num_edges > 0 && (!u || !v))) return fail(PGLAMD_E_ARG, "csr_build: NULL pointer");   // (an empty degree array has no address)
    if (!workspace || workspace_bytes < pglamd_csr_build_workspace_bytes(num_edges, num_nodes))
        return fail(PGLAMD_E_WORKSPACE, "csr_build: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t E = num_edges, N = num_nodes;
    Carver cv(workspace, workspace_bytes);
    int32_t* key_a = cv.take<int32_t>(E > 0 ? E : 1);
    int32_t* key_b = cv.take<int32_t>(E > 0 ? E : 1);
    int32_t* row_tmp = cv.take<int32_t>(E > 0 ? E : 1);
    int32_t* col_a = cv.take<int32_t>(E > 0 ? E : 1);
    int32_t* col_b = cv.take<int32_t>(E > 0 ? E : 1);
    int32_t* eid_a = cv.take<int32_t>(E > 0 ? E : 1);
    int32_t* eid_b = cv.take<int32_t>(E > 0 ? E : 1);
    uint32_t* hist = cv.take<uint32_t>(sort_hist_entries(E));
    uint32_t* totals = cv.take<uint32_t>(ceil_div((int64_t)sort_hist_entries(E), (int64_t)kScanPiece) + 1);
    uint32_t* dbase = cv.take<uint32_t>((size_t)1 << kSortMaxBits);
    uint32_t* sweep_base = cv.take<uint32_t>(((size_t)4 << kSortMaxBits) + 16);
    const size_t small_words = sweep_small_words(E);
    uint32_t* sweep_hist = cv.take<uint32_t>(((size_t)4 << kSortMaxBits) + 16 + small_words);      // [4 passes' bins] + 4 tickets (+ small builds: every pass's look-back words)
    if (!cv.ok()) return fail(PGLAMD_E_WORKSPACE, "csr_build: workspace carve overflow");
    int32_t* rows = row32 ? row32 : row_tmp;

    if (E > 0) {
        int width[8];
        const int passes = sort_plan(key_bits(N), width);
        SortArgs a{};
        a.u = u; a.us = u_stride; a.v = v; a.vs = v_stride; a.range_flag = range_flag;
        if (u_stride == 2 && v_stride == 2 && v == u + 1 && reinterpret_cast<uintptr_t>(u) % 16 == 0) a.pair = 1;
        else if (u_stride == 2 && v_stride == 2 && u == v + 1 && reinterpret_cast<uintptr_t>(v) % 16 == 0) a.pair = 2;
        a.hist = hist; a.dbase = dbase; a.n = E; a.n_rows = N; a.nblk = (int)ceil_div(E, (int64_t)kSortTile);
        const int group = passes <= 4 ? sweep_group(E) : 0;
        int bin0[8] = {0};
        if (group) {
            SweepHistArgs h{};
            h.u = u; h.us = u_stride; h.n = E; h.n_rows = N; h.nblk = a.nblk; h.passes = passes; h.ghist = sweep_hist;
            int sh = 0, bins = 0;
            for (int p = 0; p < passes; ++p) { h.shift[p] = sh; h.width[p] = width[p]; sh += width[p]; bin0[p] = bins; bins += 1 << width[p]; }
            PGLAMD_HIP_CHECK(hipMemsetAsync(sweep_hist, 0, (((size_t)4 << kSortMaxBits) + 16 + (small_words ? (size_t)a.nblk * bins : 0)) * sizeof(uint32_t), st));
            hipLaunchKernelGGL(sweep_hist_kernel, dim3((unsigned)(a.nblk < 1024 ? a.nblk : 1024)), dim3(kSortThreads), 0, st, h);
            PGLAMD_LAUNCH_CHECK();
            hipLaunchKernelGGL(sweep_base_kernel, dim3((unsigned)passes), dim3(kBlock), 0, st, h, sweep_base);
            PGLAMD_LAUNCH_CHECK();
            a.status = hist; a.group = group; a.clear_status = small_words ? 0 : 1;
        }
        int shift = 0;
        for (int p = 0; p < passes; ++p) {
            const bool first = p == 0, last = p == passes - 1;
            a.shift = shift;
            if (group) {
                a.dbase = sweep_base + bin0[p]; a.ticket = sweep_hist + ((size_t)4 << kSortMaxBits) + p;
                if (small_words) a.status = sweep_hist + ((size_t)4 << kSortMaxBits) + 16 + (size_t)a.nblk * bin0[p];
            }
            a.key_in = (p & 1) ? key_a : key_b; a.col_in = (p & 1) ? col_a : col_b; a.eid_in = (p & 1) ? eid_a : eid_b;      // pass 0 writes A, pass 1 reads A ...
            a.key_out = (p & 1) ? key_b : key_a; a.col_out = (p & 1) ? col_b : col_a; a.eid_out = (p & 1) ? eid_b : eid_a;
            if (last) { a.row32 = rows; a.col32 = col32; a.eid32 = eid32; a.sorted_u = sorted_u; a.sorted_v = sorted_v; a.sorted_eid = sorted_eid; }
            int32_t rc;
            if (group) {
                if (first && last) rc = sweep_pass<true, true>(width[p], a, st);
                else if (first) rc = sweep_pass<true, false>(width[p], a, st);
                else if (last) rc = sweep_pass<false, true>(width[p], a, st);
                else rc = sweep_pass<false, false>(width[p], a, st);
            } else
            if (first && last) rc = sort_pass<true, true>(width[p], a, totals, st);
            else if (first) rc = sort_pass<true, false>(width[p], a, totals, st);
            else if (last) rc = sort_pass<false, true>(width[p], a, totals, st);
            else rc = sort_pass<false, false>(width[p], a, totals, st);
            if (rc != PGLAMD_OK) return rc;
            shift += width[p];
        }
    }
    hipLaunchKernelGGL(row_bounds_kernel<int32_t>, dim3(grid_for(ceil_div(E + 1, (int64_t)kBoundsVec))), dim3(kBlock), 0, st, rows, E, N, indptr);
    PGLAMD_LAUNCH_CHECK();
    if (N > 0) {
        hipLaunchKernelGGL(degree_kernel, dim3(grid_for(N)), dim3(kBlock), 0, st, indptr, N, degree);
        PGLAMD_LAUNCH_CHECK();
    }
    return PGLAMD_OK;
}

extern "C" size_t pglamd_unique_segment_workspace_bytes(int64_t num_edges, int64_t num_nodes) {
    (void)num_edges;
    const int64_t N = num_nodes > 0 ? num_nodes : 1;
    return align_up((size_t)N * 8, 256) + align_up(scan_temp_bytes(N), 256) + 512;
}

extern "C" int32_t pglamd_unique_segment(const int64_t* degree, const int64_t* sorted_u, int64_t num_edges,
                                         int64_t num_nodes, int64_t* uniq_ind, int64_t* segment_ids,
                                         int64_t* num_uniq, void* workspace, size_t workspace_bytes, void* stream) {
    if (!degree || !uniq_ind || !num_uniq || (num_edges > 0 && (!sorted_u || !segment_ids)))
        return fail(PGLAMD_E_ARG, "unique_segment: NULL pointer");
    if (!workspace || workspace_bytes < pglamd_unique_segment_workspace_bytes(num_edges, num_nodes))
        return fail(PGLAMD_E_WORKSPACE, "unique_segment: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t N = num_nodes;
    if (N == 0) { PGLAMD_HIP_CHECK(hipMemsetAsync(num_uniq, 0, 8, st)); return PGLAMD_OK; }
    Carver cv(workspace, workspace_bytes);
    int64_t* rank = cv.take<int64_t>(N);
    size_t temp_bytes = scan_temp_bytes(N);
    void* temp = cv.take<char>(temp_bytes);
    (void)temp_bytes;
    { const int32_t rc = exclusive_scan64(NonEmpty{degree}, N, rank, temp, st); if (rc != PGLAMD_OK) return rc; }   // rank of every non-empty row
    hipLaunchKernelGGL(uniq_scatter_kernel, dim3(grid_for(N)), dim3(kBlock), 0, st, degree, rank, N, uniq_ind, num_uniq);
    PGLAMD_LAUNCH_CHECK();
    if (num_edges > 0) {
        hipLaunchKernelGGL(seg_ids_kernel, dim3(grid_for(num_edges)), dim3(kBlock), 0, st, sorted_u, rank, num_edges, segment_ids);
        PGLAMD_LAUNCH_CHECK();
    }
    return PGLAMD_OK;
}

extern "C" size_t pglamd_exclusive_scan_i64_workspace_bytes(int64_t n) { return exclusive_scan64_temp_bytes(n) + 256; }

extern "C" int32_t pglamd_exclusive_scan_i64(const int64_t* in, int64_t n, int64_t* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 0 || (n > 0 && (!in || !out))) return fail(PGLAMD_E_ARG, "exclusive_scan_i64: bad argument");
    if (n == 0) return PGLAMD_OK;
    if (!workspace || workspace_bytes < pglamd_exclusive_scan_i64_workspace_bytes(n)) return fail(PGLAMD_E_WORKSPACE, "exclusive_scan_i64: workspace too small");
    return exclusive_scan64(LoadI64{in}, n, out, workspace, static_cast<hipStream_t>(stream));
}

extern "C" int32_t pglamd_narrow_i64(const int64_t* in, int64_t in_stride, int64_t n, int32_t* out, void* stream) {
    if (n < 0 || (n > 0 && (!in || !out))) return fail(PGLAMD_E_ARG, "narrow_i64: bad argument");
    if (n == 0) return PGLAMD_OK;
    hipLaunchKernelGGL(narrow_i64_kernel, dim3(grid_for(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), in, in_stride, n, out);
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}

extern "C" int32_t pglamd_seg_ptr_from_ids(const void* ids, int32_t ids_i64, int64_t num_rows, int64_t n_seg,
                                           int64_t* seg_ptr, void* stream) {
    if (!seg_ptr || (num_rows > 0 && !ids) || num_rows < 0 || n_seg < 0) return fail(PGLAMD_E_ARG, "seg_ptr_from_ids: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (ids_i64)
        hipLaunchKernelGGL(row_bounds_kernel<int64_t>, dim3(grid_for(ceil_div(num_rows + 1, (int64_t)kBoundsVec))), dim3(kBlock), 0, st,
                           static_cast<const int64_t*>(ids), num_rows, n_seg, seg_ptr);
    else
        hipLaunchKernelGGL(row_bounds_kernel<int32_t>, dim3(grid_for(ceil_div(num_rows + 1, (int64_t)kBoundsVec))), dim3(kBlock), 0, st,
                           static_cast<const int32_t*>(ids), num_rows, n_seg, seg_ptr);
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}
