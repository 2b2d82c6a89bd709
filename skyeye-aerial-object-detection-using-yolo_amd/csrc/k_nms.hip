// Per-image non-maximum suppression on gfx950.
//
// Replaces non_max_suppression (reference skyeye/utils/metrics.py:361-457) including the greedy suppression it
// delegates to torchvision.ops.nms (call site metrics.py:442; semantics restated in oracle/sky_oracle_nms.c).
//
//   1. count    per 256-row block: how many candidate entries the rows produce   (metrics.py:389,402-422,425-426)
//   2. (scan)   every emit block adds up the counts of the blocks before it (deterministic, ordered compaction - no atomics)
//   3. emit     candidate records in (row, class) order + 64-bit sort keys
//               key = ~monotone(score) << 32 | candidate index  => ascending key order is "score descending,
//               lower index first", the tie rule fixed by this build
//   4. sort     per image, padded to a power of two: bitonic sort of chunks of 16384 keys in LDS, then merge levels (merge path,
//               16384 -> 32768 -> ...) between two key buffers; an image with at most 16384 candidates takes none of them
//   5. greedy   one workgroup of 8 wavefronts per image walks the sorted candidates 512 at a time: wave by wave the 64 lanes
//               resolve among themselves in score order with ballots, the later waves test their boxes against the boxes that
//               were just kept (LDS); stops at max_det kept or max_nms visited (metrics.py:434-435,443-444)
//
// This file is compiled with -ffp-contract=off: IoU and the class-offset arithmetic must round exactly like the
// fp32 tensor ops of the reference so that the kept set is bit-identical to the oracle's.
#include "sky_kernels.h"

#include <math.h>

#include <algorithm>

namespace sky {

static constexpr int ROWS = 256;        // rows per count/emit block
static constexpr int CHUNK = 16384;     // keys per LDS sort block (128 KiB of the CU's 160 KiB LDS)
static constexpr int SORT_T = 1024;     // threads per sort block

__device__ __forceinline__ unsigned int score_key(float s)
{
    unsigned int u = __float_as_uint(s);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending float order
    return ~u;                                        // descending
}

__device__ __forceinline__ bool class_ok(const NmsArgs& a, float c5)
{
    if (a.n_classes <= 0) return true;
    bool hit = false;
    for (int c = 0; c < a.n_classes; ++c) hit |= (c5 == (float)a.classes[c]);
    return hit;
}

// Enumerate the candidate entries of one prediction row in the reference's order; f(score, c5, c6).
template <typename F>
__device__ __forceinline__ void row_entries(const NmsArgs& a, const float* __restrict__ p, F&& f)
{
    const int nc = a.nc;
    const float obj = p[4];
    if (!(obj > a.conf)) return;                                   // metrics.py:389,402
    if (a.mode == 2) {                                             // rows that are already boxes: x1, y1, x2, y2, conf, cls
        if (class_ok(a, p[5])) f(obj, p[5], 0.0f);                 // (second stage of the tiled path: survivors of the per-tile NMS)
        return;
    }
    if (a.mode == 0) {
        if (nc > 1) {
            if (a.multi_label) {                                   // metrics.py:412-413
                for (int j = 0; j < nc; ++j) {
                    const float c = p[5 + j];
                    if (c > a.conf && class_ok(a, c)) f(obj, c, (float)j);
                }
            } else {                                               // metrics.py:416-417 (first maximal index)
                int bj = 0;
                float bc = p[5];
                for (int j = 1; j < nc; ++j) {
                    const float c = p[5 + j];
                    if (c > bc) { bc = c; bj = j; }
                }
                if (bc > a.conf && class_ok(a, bc)) f(obj, bc, (float)bj);
            }
        } else {                                                   // metrics.py:420-422
            if (class_ok(a, 0.0f)) f(obj, 0.0f, 0.0f);
        }
    } else {                                                       // corrected: conf = obj * cls, class id in col 5
        if (a.multi_label && nc > 1) {
            for (int j = 0; j < nc; ++j) {
                const float c = p[5 + j] * obj;
                if (c > a.conf && class_ok(a, (float)j)) f(c, (float)j, 0.0f);
            }
        } else {
            int bj = 0;
            float bc = p[5] * obj;
            for (int j = 1; j < nc; ++j) {
                const float c = p[5 + j] * obj;
                if (c > bc) { bc = c; bj = j; }
            }
            if (bc > a.conf && class_ok(a, (float)bj)) f(bc, (float)bj, 0.0f);
        }
    }
}

__global__ void __launch_bounds__(ROWS) nms_count_kernel(const NmsArgs a, int nblk)
{
    const int b = blockIdx.y, blk = blockIdx.x;
    const int i = blk * ROWS + threadIdx.x;
    int cnt = 0;
    if (i < a.N) row_entries(a, a.det + ((long)b * a.N + i) * (a.nc + 5), [&](float, float, float) { ++cnt; });
    __shared__ int red[ROWS / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int w = 0; w < ROWS / 64; ++w) s += red[w];
        a.blk_counts[(long)b * nblk + blk] = s;
    }
}

// Candidate records + sort keys of one 256-row block, in (row, class) order.  The block's first slot = the sum of the counts of the
// blocks before it (every block adds them up itself: at most a few hundred integers from L2, cheaper than a scan launch); block 0
// also publishes the image's total.  Inside the block: wave scans by shuffles, wave offsets through LDS.
__global__ void __launch_bounds__(ROWS) nms_emit_kernel(const NmsArgs a, int nblk)
{
    const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blk * ROWS + tid;
    const float* p = a.det + ((long)b * a.N + i) * (a.nc + 5);
    int cnt = 0;
    if (i < a.N) row_entries(a, p, [&](float, float, float) { ++cnt; });
    __shared__ int red[3][ROWS / 64];
    // counts of the blocks before this one (and, in block 0, of all blocks)
    const int* c = a.blk_counts + (long)b * nblk;
    int pre = 0, tot = 0;
    for (int j = tid; j < nblk; j += ROWS) {
        const int v = c[j];
        tot += v;
        pre += j < blk ? v : 0;
    }
    // inclusive scan of cnt over the wave
    int inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { pre += __shfl_xor(pre, o); tot += __shfl_xor(tot, o); }
    if (lane == 63) red[0][wave] = inc;
    if (lane == 0) { red[1][wave] = pre; red[2][wave] = tot; }
    __syncthreads();
    int base = 0, woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < ROWS / 64; ++w) {
        base += red[1][w];
        total += red[2][w];
        woff += w < wave ? red[0][w] : 0;
    }
    if (blk == 0 && tid == 0) a.totals[b] = total;
    long k = (long)base + woff + (inc - cnt);
    if (i < a.N && cnt) {
        row_entries(a, p, [&](float score, float c5, float c6) {
            if (k < a.cap) {
                a.keys[(long)b * a.cap + k] = ((unsigned long long)score_key(score) << 32) | (unsigned int)k;
                float* r = a.cand + ((long)b * a.cap + k) * 4;
                r[0] = score; r[1] = c5; r[2] = c6; r[3] = __int_as_float(i);
            }
            ++k;
        });
    }
}

__device__ __forceinline__ long padded_len(int total)
{
    long p = 1;
    while (p < total) p <<= 1;
    return p;
}

// Bitonic sort of one chunk of CHUNK keys inside LDS, ascending (the direction of a compare-exchange comes from the element's index
// INSIDE the chunk, so every chunk ends ascending: the merge levels below want sorted runs, not a bitonic sequence).
__global__ void __launch_bounds__(SORT_T) nms_sort_lds_kernel(const NmsArgs a)
{
    const int b = blockIdx.y;
    const long P = padded_len(a.totals[b]);
    const long base = (long)blockIdx.x * CHUNK;
    if (base >= P) return;
    const long len = P < CHUNK ? P : CHUNK;
    extern __shared__ unsigned long long sk[];   // CHUNK keys
    unsigned long long* g = a.keys + (long)b * a.cap + base;
    const long total = a.totals[b];
    for (int i = threadIdx.x; i < len; i += SORT_T) sk[i] = base + i < total ? g[i] : ~0ull;       // keys [total, P) = ~0: they sort last
    __syncthreads();
    const int half = (int)(len >> 1);
    for (long k = 2; k <= len; k <<= 1) {
        // j is a power of two: pair index t -> lo = (t / j) * 2j + t % j with shifts (the 64-bit divisions of the plain form cost more
        // than the compare-exchange)
        for (int j = (int)(k >> 1), sh = 31 - __builtin_clz((unsigned)(k >> 1)); j > 0; j >>= 1, --sh) {
            for (int t = threadIdx.x; t < half; t += SORT_T) {
                const int lo = ((t >> sh) << (sh + 1)) | (t & (j - 1));
                const int hi = lo + j;
                const bool asc = ((lo & k) == 0);
                const unsigned long long x = sk[lo], y = sk[hi];
                if ((x > y) == asc) { sk[lo] = y; sk[hi] = x; }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < len; i += SORT_T) g[i] = sk[i];
}

// Merge level: sorted runs of R keys -> sorted runs of 2R keys, from `src` into `dst` (the two key buffers alternate level by level;
// an image whose padded length P is at most R is finished and is left where it is: nms_sorted_keys() tells the reader where).  A
// workgroup produces MSEG * 256 consecutive output keys of one pair of runs: two binary searches along the segment's first and last
// diagonal (merge path: i keys of run A and d - i of run B lie below the cut; on ties B goes first) bound the pieces of A and B it
// needs, those go to LDS with coalesced loads, and every thread cuts its own MSEG keys out of the LDS pieces the same way.
static constexpr int MSEG = 8, MWG = MSEG * 256;
__device__ __forceinline__ long merge_cut(const unsigned long long* A, long na, const unsigned long long* Bk, long nb, long d)
{
    long lo = d > nb ? d - nb : 0, hi = d < na ? d : na;            // i in [lo, hi]: keys taken from A among the first d
    while (lo < hi) {
        const long i = (lo + hi) >> 1;                              // A[i] against B[d - i - 1]
        if (A[i] < Bk[d - i - 1]) lo = i + 1; else hi = i;
    }
    return lo;
}
__global__ void __launch_bounds__(256) nms_merge_kernel(const NmsArgs a, long R, int level)
{
    __shared__ unsigned long long sa[MWG], sb[MWG];
    __shared__ long cut[2];
    const int b = blockIdx.y, tid = threadIdx.x;
    const long P = padded_len(a.totals[b]);
    if (P <= R) return;
    const unsigned long long* src = ((level & 1) ? a.keys : a.keys2) + (long)b * a.cap;
    unsigned long long* dst = ((level & 1) ? a.keys2 : a.keys) + (long)b * a.cap;
    for (long o0 = (long)blockIdx.x * MWG; o0 < P; o0 += (long)gridDim.x * MWG) {
        const long pair = o0 / (2 * R), d0 = o0 - pair * 2 * R;     // output keys [d0, d0 + MWG) of this pair of runs
        const unsigned long long* A = src + pair * 2 * R;
        const unsigned long long* Bk = A + R;
        if (tid < 2) cut[tid] = merge_cut(A, R, Bk, R, d0 + tid * MWG);
        __syncthreads();
        const long ia0 = cut[0], ia1 = cut[1], ib0 = d0 - ia0, ib1 = d0 + MWG - ia1;
        const int na = (int)(ia1 - ia0), nb = (int)(ib1 - ib0);     // na + nb = MWG
        for (int i = tid; i < na; i += 256) sa[i] = A[ia0 + i];
        for (int i = tid; i < nb; i += 256) sb[i] = Bk[ib0 + i];
        __syncthreads();
        const int d = tid * MSEG;
        int i = (int)merge_cut(sa, na, sb, nb, d), j = d - i;
        unsigned long long outv[MSEG];
#pragma unroll
        for (int e = 0; e < MSEG; ++e) {
            const bool ta = j >= nb || (i < na && sa[i] < sb[j]);
            outv[e] = ta ? sa[i] : sb[j];
            i += ta ? 1 : 0;
            j += ta ? 0 : 1;
        }
#pragma unroll
        for (int e = 0; e < MSEG; ++e) dst[o0 + d + e] = outv[e];
        __syncthreads();                                            // the LDS pieces are rewritten by the next segment
    }
}

// where the sorted keys of image b ended up: every merge level an image takes flips the buffer
__device__ __forceinline__ const unsigned long long* nms_sorted_keys(const NmsArgs& a, int b)
{
    const long P = padded_len(a.totals[b]);
    int levels = 0;
    for (long R = CHUNK; R < P; R <<= 1) ++levels;
    return ((levels & 1) ? a.keys2 : a.keys) + (long)b * a.cap;
}

__device__ __forceinline__ bool iou_gt(float kx1, float ky1, float kx2, float ky2, float karea, float x1, float y1, float x2,
                                       float y2, float area, float thr)
{
    float w = fminf(kx2, x2) - fmaxf(kx1, x1);
    float h = fminf(ky2, y2) - fmaxf(ky1, y1);
    w = w > 0.0f ? w : 0.0f;
    h = h > 0.0f ? h : 0.0f;
    const float inter = w * h;
    const float iou = inter / (karea + area - inter);
    return iou > thr;
}

// Filter in front of iou_gt: boxes that do not overlap in both axes have inter = 0 and are never suppressed (0 / union > thr is false
// for every union, NaN included, as long as thr >= 0) -- most pairs, since the class offset moves the classes apart.  Same
// subtractions as iou_gt, so the filter passes exactly the pairs whose w and h are positive there.
__device__ __forceinline__ bool boxes_overlap(float kx1, float ky1, float kx2, float ky2, float x1, float y1, float x2, float y2)
{
    const float w = fminf(kx2, x2) - fmaxf(kx1, x1);
    const float h = fminf(ky2, y2) - fmaxf(ky1, y1);
    return w > 0.0f && h > 0.0f;
}

// One workgroup of GW wavefronts per image.  The sorted candidates are walked GW * 64 at a time, one per thread; inside such a
// super-block wave w owns candidates [64 w, 64 w + 64).  In block order: the owning wave resolves its 64 lanes among themselves in
// score order with ballots (appending to the kept list in LDS), a barrier publishes the new kept boxes, and every LATER wave tests
// its own candidates against just those -- so the n x kept IoU tests of the greedy scan run GW waves wide while the order of
// decisions (and with it the kept set) is exactly that of the sequential scan.  Stops at max_det kept or max_nms visited
// (metrics.py:434-435,443-444).
static constexpr int GW = 8;
__global__ void __launch_bounds__(GW * 64) nms_greedy_kernel(const NmsArgs a)
{
    extern __shared__ float kb[];   // [max_det][5]: x1, y1, x2, y2, area of kept boxes (class offset applied); then int s_kept[2]
    int* const s_kept = reinterpret_cast<int*>(kb + (size_t)a.max_det * 5);
    float* const kbb = reinterpret_cast<float*>(s_kept + 2);      // [64][5]: the boxes of the block being resolved
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int no = a.nc + 5;
    const int cols = (a.mode == 0 && a.nc > 1) ? 7 : 6;
    int n = a.totals[b];
    if (n > a.max_nms) n = a.max_nms;
    if ((long)n > a.cap) n = (int)a.cap;
    int kept = 0;
    const bool nofilter = !(a.iou >= 0.0f);          // a negative threshold suppresses disjoint boxes too: no overlap filter
    const unsigned long long* keys = nms_sorted_keys(a, b);
    float* out = a.out + (long)b * a.out_stride;
    for (int sb = 0; sb < n && kept < a.max_det; sb += GW * 64) {
        const int k = sb + tid;
        bool alive = k < n;
        float b0 = 0, b1 = 0, b2 = 0, b3 = 0, score = 0, c5 = 0, c6 = 0;
        float x1 = 0, y1 = 0, x2 = 0, y2 = 0, area = 0;
        if (alive) {
            const unsigned int ci = (unsigned int)(keys[k] & 0xffffffffull);
            const float* r = a.cand + ((long)b * a.cap + ci) * 4;
            score = r[0]; c5 = r[1]; c6 = r[2];
            const float* p = a.det + ((long)b * a.N + __float_as_int(r[3])) * no;
            if (a.mode == 0 || a.mode == 2) { b0 = p[0]; b1 = p[1]; b2 = p[2]; b3 = p[3]; }
            else {
                const float hw = p[2] / 2.0f, hh = p[3] / 2.0f;
                b0 = p[0] - hw; b1 = p[1] - hh; b2 = p[0] + hw; b3 = p[1] + hh;
            }
            const float c = a.agnostic ? c5 * 0.0f : c5 * a.max_wh;    // metrics.py:438
            x1 = b0 + c; y1 = b1 + c; x2 = b2 + c; y2 = b3 + c;
            area = (x2 - x1) * (y2 - y1);
        }
        // This thread's box against up to 64 boxes in LDS ([cnt][5]): the boxes among `allowed` that suppress it.  Pass 1 marks the
        // overlapping ones (branch-free, the LDS reads pipeline), pass 2 takes the IoU decision for the marked ones only (per-lane
        // lists, a handful of entries).
        auto suppressors = [&](const float* boxes, int cnt, unsigned long long allowed) -> unsigned long long {
            unsigned long long mask = 0ull;
            if (nofilter) mask = cnt == 64 ? ~0ull : (1ull << cnt) - 1ull;
            else {
#pragma unroll 4
                for (int t = 0; t < cnt; ++t) {
                    const float* q = boxes + t * 5;
                    mask |= (unsigned long long)boxes_overlap(q[0], q[1], q[2], q[3], x1, y1, x2, y2) << t;
                }
            }
            mask &= allowed;
            unsigned long long hits = 0ull;
            while (__any(mask != 0ull)) {
                if (mask) {
                    const int t = __ffsll((long long)mask) - 1;
                    mask &= mask - 1ull;
                    const float* q = boxes + t * 5;
                    if (iou_gt(q[0], q[1], q[2], q[3], q[4], x1, y1, x2, y2, area, a.iou)) hits |= 1ull << t;
                }
            }
            return hits;
        };
        auto test_kept = [&](int t0, int t1) {
            for (int tb = t0; tb < t1; tb += 64)
                if (suppressors(kb + tb * 5, tb + 64 < t1 ? 64 : t1 - tb, alive ? ~0ull : 0ull)) alive = false;
        };
        // against the boxes kept in earlier super-blocks
        if (kept > 0 && __any(alive)) test_kept(0, kept);
        for (int blk = 0; blk < GW; ++blk) {
            if (sb + blk * 64 >= n) break;                       // (uniform)
            const int kept0 = kept;
            if (wave == blk) {
                // Among the 64 lanes, in score order, without a serial walk: (1) every lane collects `sup`, the earlier alive lanes whose
                // box suppresses its own (the block's boxes go through an LDS scratch, then the same two passes as against the kept
                // list); (2) the kept set K is the unique set with
                // "lane j is in K <=> alive and no lane of sup_j is in K".  Iterating K <- {j alive : sup_j & K == 0} from K = all alive
                // settles lane 0 first, then every lane whose earlier lanes are settled: it reaches that set (in practice in 2-3 rounds).
                const unsigned long long A = __ballot(alive);
                {
                    float* q = kbb + lane * 5;
                    q[0] = x1; q[1] = y1; q[2] = x2; q[3] = y2; q[4] = area;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");           // (one wave writes and reads kbb: no barrier)
                __builtin_amdgcn_wave_barrier();
                const unsigned long long sup = suppressors(kbb, 64, alive ? A & ((1ull << lane) - 1ull) : 0ull);
                unsigned long long K = A, prev;
                do {
                    prev = K;
                    K = __ballot(alive && (sup & prev) == 0ull);
                } while (K != prev);
                // at most max_det boxes in all: the first (max_det - kept) lanes of K
                const int rank = __popcll(K & ((1ull << lane) - 1ull)), avail = a.max_det - kept;
                const bool mine = ((K >> lane) & 1ull) != 0ull && rank < avail;
                const int my = mine ? kept + rank : -1;
                const int nk = __popcll(K);
                const int kk = kept + (nk < avail ? nk : avail);
                if (my >= 0) {
                    float* q = kb + my * 5;
                    q[0] = x1; q[1] = y1; q[2] = x2; q[3] = y2; q[4] = area;
                    float* o = out + (long)my * 7;
                    o[0] = b0; o[1] = b1; o[2] = b2; o[3] = b3; o[4] = score; o[5] = c5;
                    o[6] = cols == 7 ? c6 : 0.0f;
                }
                if (lane == 0) s_kept[blk & 1] = kk;
            }
            __syncthreads();                                     // the kept list up to s_kept is published
            kept = s_kept[blk & 1];                              // (slot blk & 1 is next written two barriers from here)
            if (kept >= a.max_det) break;                        // (uniform)
            if (wave > blk && __any(alive)) test_kept(kept0, kept);
        }
        __syncthreads();   // the s_kept slots and the waves' roles start over
    }
    if (tid == 0) a.counts[(long)b * a.counts_stride] = kept;
    // rows past the kept ones are defined (zeros): callers hand over uninitialised buffers and the fixed-capacity block
    // that travels through the RCCL all-gather is the same bytes on every run
    for (int i = kept * 7 + tid; i < a.max_det * 7; i += GW * 64) out[i] = 0.0f;
}

static long next_pow2(long v)
{
    long p = 1;
    while (p < v) p <<= 1;
    return p;
}

size_t nms_workspace_bytes(int B, int N, int nc, int multi_label, long* cap_out)
{
    const long cap = next_pow2((long)N * ((multi_label && nc > 1) ? nc : 1));
    const int nblk = (N + ROWS - 1) / ROWS;
    if (cap_out) *cap_out = cap;
    size_t bytes = 0;
    bytes += ((size_t)B * nblk * sizeof(int) + 255) / 256 * 256;
    bytes += ((size_t)B * sizeof(int) + 255) / 256 * 256;
    bytes += (size_t)2 * B * cap * sizeof(unsigned long long);
    bytes += (size_t)B * cap * 4 * sizeof(float);
    return bytes;
}

hipError_t launch_nms(const NmsArgs& a, hipStream_t s)
{
    const int nblk = (a.N + ROWS - 1) / ROWS;
    hipLaunchKernelGGL(nms_count_kernel, dim3(nblk, a.B), dim3(ROWS), 0, s, a, nblk);
    hipLaunchKernelGGL(nms_emit_kernel, dim3(nblk, a.B), dim3(ROWS), 0, s, a, nblk);
    const unsigned chunks = (unsigned)((a.cap + CHUNK - 1) / CHUNK);
    static size_t attr[16] = {0};
    {
        const hipError_t e = ensure_lds_attr(reinterpret_cast<const void*>(nms_sort_lds_kernel), CHUNK * 8, a.device, attr);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(nms_sort_lds_kernel, dim3(chunks, a.B), dim3(SORT_T), CHUNK * 8, s, a);
    {
        int level = 1;
        for (long R = CHUNK; R < a.cap; R <<= 1, ++level)
            hipLaunchKernelGGL(nms_merge_kernel, dim3((unsigned)std::min<long>((a.cap + MWG - 1) / MWG, 64), a.B), dim3(256), 0, s, a, R, level);
    }
    const size_t glds = (size_t)a.max_det * 5 * sizeof(float) + 2 * sizeof(int) + 64 * 5 * sizeof(float);
    if (glds > 48 * 1024) {         // max_detections above ~2400: the kept list needs more than the default dynamic LDS limit
        static size_t gattr[16] = {0};
        const hipError_t e = ensure_lds_attr(reinterpret_cast<const void*>(nms_greedy_kernel), glds, a.device, gattr);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(nms_greedy_kernel, dim3(a.B), dim3(GW * 64), glds, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ box_iou
// Pairwise IoU of evaluation accounting (reference metrics.py:17-44, used by validate.py:process_batch :71-108):
// inter = clamp(min(x2) - max(x1), 0) * clamp(min(y2) - max(y1), 0); heights carry + 1e-7, the union one more 1e-7;
// the same operation order as the tensor expression ((w1*h1 + w2*h2) - inter) + 1e-7, no FMA contraction (this file).
// box1 is [4, n] (the file's literal indexing, box1_4xn = 1) or [n, 4]; box2 is [m, 4]; out is [n, m].
__global__ void box_iou_kernel(const float* __restrict__ b1, int n, int b1_4xn, const float* __restrict__ b2, int m, float* __restrict__ out)
{
    const long total = (long)n * m;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int i = (int)(idx / m), j = (int)(idx - (long)i * m);
        float ax1, ay1, ax2, ay2;
        if (b1_4xn) { ax1 = b1[i]; ay1 = b1[n + i]; ax2 = b1[2 * n + i]; ay2 = b1[3 * n + i]; }
        else { ax1 = b1[4 * i]; ay1 = b1[4 * i + 1]; ax2 = b1[4 * i + 2]; ay2 = b1[4 * i + 3]; }
        const float bx1 = b2[4 * j], by1 = b2[4 * j + 1], bx2 = b2[4 * j + 2], by2 = b2[4 * j + 3];
        float iw = fminf(ax2, bx2) - fmaxf(ax1, bx1);
        float ih = fminf(ay2, by2) - fmaxf(ay1, by1);
        iw = iw > 0.0f ? iw : 0.0f;
        ih = ih > 0.0f ? ih : 0.0f;
        const float inter = iw * ih;
        const float w1 = ax2 - ax1, h1 = ay2 - ay1 + 1e-7f;
        const float w2 = bx2 - bx1, h2 = by2 - by1 + 1e-7f;
        const float uni = (w1 * h1) + (w2 * h2) - inter + 1e-7f;
        out[idx] = inter / uni;
    }
}

hipError_t launch_box_iou(const float* box1, int n, int box1_4xn, const float* box2, int m, float* out, hipStream_t s)
{
    const long total = (long)n * m;
    if (total == 0) return hipSuccess;
    long grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(box_iou_kernel, dim3((unsigned)grid), dim3(256), 0, s, box1, n, box1_4xn, box2, m, out);
    return hipGetLastError();
}

}  // namespace sky
