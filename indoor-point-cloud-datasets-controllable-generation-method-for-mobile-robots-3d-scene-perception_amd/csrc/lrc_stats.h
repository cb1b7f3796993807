// lrc_stats.h -- per-frame mean and standard deviation exactly as numpy computes them (gfx950 only).
// Included by lidarcast.hip inside its anonymous namespace.
//
// What it replaces: the np.mean / np.std calls over every frame's ranges (and incident angles) in the reference's
// scan loop (s3dis_simulator.py:276-286) -- after the scan moved to the GPU they were what S3DISSimulator.run_simulation
// spent its time on (64 frames x 65 k values: 5 of 8 ms).  The statistics are part of the frames a drop-in must
// reproduce, so the kernel follows numpy's arithmetic operation by operation (numpy 2.2, default buffer size):
//   np.add.reduce over a contiguous 1-D array of n values of type T (no casting):
//     the array is consumed in buffer chunks of 8192 elements; chunk sums are added left to right;
//     a chunk is summed by pairwise_sum (numpy/_core/src/umath/loops_utils.h.src):
//       n < 8    : res = 0; res += a[i] in order
//       n <= 128 : r[j] = a[j] (j < 8); r[j] += a[i + j] for i = 8, 16, ... < n - n % 8;
//                  res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)); res += a[i] for the n % 8 tail
//       else     : n2 = n / 2; n2 -= n2 % 8; pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2)
//   np.mean = sum / n in T;  np.std = sqrt(sum((a - mean)^2) / n), every operation in T, mean = sum / n as above
// (tests/test_oracle.py checks this restatement against numpy for many lengths; the -m gpu tests check the kernel).
// A full chunk (8192 = 64 leaves of 128) is a balanced tree and runs in parallel: 8 lanes per leaf carry the eight
// accumulators, three xor-shuffles combine them in numpy's bracket order (IEEE addition commutes, so which lane of a
// pair holds which operand is immaterial), six LDS levels combine the leaves.  The trailing partial chunk has an
// irregular tree, but all its leaves sit on two adjacent depths, so its nodes at the shallower of the two are reached
// by index arithmetic alone and reduce as a balanced tree again (block_reduce_numpy).
#pragma once

template <typename T, bool SQ>
__device__ __forceinline__ T stat_term(const T* a, uint64_t i, T mean) {
    if (!SQ) return a[i];
    const T x = a[i] - mean;          // two roundings, as numpy's (arr - mean) then x * x (no FMA: -ffp-contract=off)
    return x * x;
}

// one leaf (n <= 128) by 8 lanes: numpy's 8 accumulators, bracketed combine, sequential tail; valid in every lane of
// the group.  All of the leaf's rows are fetched before the first add (the adds are a dependent chain, the loads need
// not be), rows beyond the leaf are neither fetched nor added.
template <typename T, bool SQ>
__device__ __forceinline__ T leaf_sum8(const T* a, uint32_t n, uint32_t j, T mean) {
    if (n < 8u) {                                   // res = 0; res += a[i]
        T res = (T)0;
        for (uint32_t i = 0; i < n; ++i) res += stat_term<T, SQ>(a, i, mean);
        return res;
    }
    const uint32_t rows = n / 8u;                   // 1..16 full rows of 8
    T v[16];
#pragma unroll
    for (uint32_t k = 0; k < 16u; ++k) v[k] = k < rows ? stat_term<T, SQ>(a, k * 8u + j, mean) : (T)0;
    T r = v[0];
#pragma unroll
    for (uint32_t k = 1; k < 16u; ++k)
        if (k < rows) r += v[k];
    r = r + __shfl_xor(r, 1, 64);
    r = r + __shfl_xor(r, 2, 64);
    r = r + __shfl_xor(r, 4, 64);
    for (uint32_t i = rows * 8u; i < n; ++i) r += stat_term<T, SQ>(a, i, mean);      // every lane adds the same tail
    return r;
}

constexpr uint32_t kStatBatch = 8;       // full 8192-element chunks whose leaves are summed side by side

// np.add.reduce(a[0..n)) (or of (a - mean)^2) by one 256-thread workgroup; the result is valid in thread 0
template <typename T, bool SQ>
__device__ T block_reduce_numpy(const T* a, uint64_t n, T mean, T* s_leaf /* 64 * kStatBatch */) {
    const uint32_t tid = threadIdx.x;
    const uint32_t grp = tid >> 3, j = tid & 7u;       // 32 groups of 8 lanes
    T total = (T)0;
    const uint64_t full = n / 8192;
    for (uint64_t c0 = 0; c0 < full; c0 += kStatBatch) {
        const uint32_t nb = (uint32_t)(full - c0 < kStatBatch ? full - c0 : kStatBatch);
        const T* ch = a + c0 * 8192;
        for (uint32_t leaf = grp; leaf < nb * 64u; leaf += 32u) {          // leaves of all chunks of the batch
            const T r = leaf_sum8<T, SQ>(ch + (uint64_t)leaf * 128u, 128u, j, mean);
            if (j == 0) s_leaf[leaf] = r;
        }
        __syncthreads();
        for (uint32_t w = 32; w >= 1; w >>= 1) {          // per chunk a balanced tree: node k = children 2k, 2k + 1
            const uint32_t chunk = tid / w, k = tid - chunk * w;
            T v = (T)0;
            if (chunk < nb) v = s_leaf[chunk * 64u + 2u * k] + s_leaf[chunk * 64u + 2u * k + 1u];
            __syncthreads();
            if (chunk < nb) s_leaf[chunk * 64u + k] = v;
            __syncthreads();
        }
        if (tid == 0)
            for (uint32_t c = 0; c < nb; ++c) total = (c0 + c == 0) ? s_leaf[c * 64u] : total + s_leaf[c * 64u];
        __syncthreads();
    }
    const uint32_t rest = (uint32_t)(n - full * 8192);
    if (rest) {
        // The ragged tail chunk.  pairwise_sum halves (n2 = n/2 - (n/2) % 8) until a piece has <= 128 elements; sibling
        // sizes differ by < 16, so all leaves sit on two adjacent depths D-1 and D and the tree is complete down to depth
        // E = D - 1 (verified against the literal recursion for every length 1..8191, tests/test_numpy_reduction_model.py).
        // Slot s of the 2^E nodes at depth E is reached by following the bits of s from the root -- registers only --
        // and is either a leaf or the sum of its two leaf children; the 2^E values then reduce as a balanced tree.
        const T* ch = a + full * 8192;
        T part = (T)0;
        if (rest <= 128u) {
            part = leaf_sum8<T, SQ>(ch, rest, j, mean);         // every thread computes it; thread 0's copy is used
        } else {
            uint32_t size = rest, D = 0;
            while (size > 128u) { uint32_t h = size / 2u; h -= h % 8u; size -= h; ++D; }      // follow the larger child
            const uint32_t E = D - 1u, S = 1u << E;                                            // S <= 64
            for (uint32_t slot = grp; slot < S; slot += 32u) {
                uint32_t st = 0, m = rest;
                for (uint32_t lvl = 0; lvl < E; ++lvl) {
                    uint32_t h = m / 2u; h -= h % 8u;
                    if ((slot >> (E - 1u - lvl)) & 1u) { st += h; m -= h; } else m = h;
                }
                T v;
                if (m <= 128u) v = leaf_sum8<T, SQ>(ch + st, m, j, mean);
                else {
                    uint32_t h = m / 2u; h -= h % 8u;
                    const T l = leaf_sum8<T, SQ>(ch + st, h, j, mean);
                    const T r = leaf_sum8<T, SQ>(ch + st + h, m - h, j, mean);
                    v = l + r;
                }
                if (j == 0) s_leaf[slot] = v;
            }
            __syncthreads();
            for (uint32_t w = S >> 1; w >= 1; w >>= 1) {
                T v = (T)0;
                if (tid < w) v = s_leaf[2u * tid] + s_leaf[2u * tid + 1u];
                __syncthreads();
                if (tid < w) s_leaf[tid] = v;
                __syncthreads();
            }
            part = s_leaf[0];
            __syncthreads();
        }
        if (tid == 0) total = full == 0 ? part : total + part;
    }
    return total;
}

// ---- mean and population standard deviation per segment (frame), numpy's way, one workgroup per 8192-element CHUNK ----------
// np.add.reduce adds the pairwise sums of the buffer chunks left to right, so the chunks of all frames are independent: kStatPar
// workgroups per frame stride over its chunks (64 frames of ~65 k values: 512 workgroups instead of the 64 the first version
// ran -- 310 us, a hundredth of the memory rate), then one thread per frame adds its chunk sums in chunk order, divides, and
// the same again for the squares.  Chunk c of segment s (first row start_s) lands in partial[start_s / 8192 + s + c]: the
// ranges of different segments never overlap (floor(a + b) >= floor(a) + floor(b)), total rows / 8192 + segments + 1 values.
constexpr uint32_t kStatPar = 8;

__device__ __forceinline__ uint64_t segment_start(const uint64_t* counts, uint64_t first_row, uint64_t seg) {
    uint64_t start = first_row;                      // a few hundred counts at most, L2 resident
    for (uint64_t k = 0; k < seg; ++k) start += counts[k];
    return start;
}

template <typename T, bool SQ>
__global__ __launch_bounds__(256) void segment_chunk_sums_kernel(const T* values, const uint64_t* counts, uint64_t first_row,
                                                                 uint64_t num_segments, const T* mean, T* partial) {
    __shared__ T s_leaf[64 * kStatBatch];
    const uint64_t seg = blockIdx.x / kStatPar;
    if (seg >= num_segments) return;
    const uint64_t n = counts[seg];
    const uint64_t start = segment_start(counts, first_row, seg);
    T* out = partial + (start - first_row) / 8192 + seg;
    const T m = SQ ? mean[seg] : (T)0;
    for (uint64_t c = blockIdx.x - seg * kStatPar; c * 8192 < n; c += kStatPar) {
        uint64_t len = n - c * 8192;
        if (len > 8192) len = 8192;
        // the length is made opaque: with the range [1, 8192] visible (and -fno-honor-nans, the library's device flag) this
        // compiler drops the ragged chunk's result for T = double (tools/micro/stats_check.hip shows it); the one-workgroup
        // form of round 2, whose length came straight from memory, never met the problem
        asm volatile("" : "+s"(len));
        const T sum = block_reduce_numpy<T, SQ>(values + start + c * 8192, len, m, s_leaf);
        if (threadIdx.x == 0) out[c] = sum;
        __syncthreads();
    }
}

// SQ = false: out[seg] = mean = (chunk sums added left to right) / n.  SQ = true: out[seg] = sqrt(that sum / n).
template <typename T, bool SQ>
__global__ __launch_bounds__(256) void segment_combine_kernel(const T* partial, const uint64_t* counts, uint64_t first_row,
                                                              uint64_t num_segments, T* out) {
    const uint64_t seg = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (seg >= num_segments) return;
    const uint64_t n = counts[seg];
    if (n == 0) { out[seg] = (T)0; return; }
    const uint64_t nch = (n + 8191) / 8192;
    const T* p = partial + (segment_start(counts, first_row, seg) - first_row) / 8192 + seg;
    T total = p[0];
    for (uint64_t c = 1; c < nch; ++c) total = total + p[c];
    const T q = total / (T)n;
    if (!SQ) out[seg] = q;
    else out[seg] = sizeof(T) == 4 ? (T)__builtin_sqrtf((float)q) : (T)__builtin_sqrt((double)q);
}

// values of scratch a call over `rows` rows in `num_segments` segments needs
inline uint64_t segment_stats_scratch_values(uint64_t num_segments, uint64_t rows) { return rows / 8192 + num_segments + 2; }

// four launches on `st`; `partial`: segment_stats_scratch_values(...) values of scratch the caller owns on that stream
template <typename T>
void launch_segment_stats(hipStream_t st, const T* values, const uint64_t* counts, uint64_t first_row, uint64_t num_segments,
                          T* partial, T* out_mean, T* out_std) {
    if (num_segments == 0) return;
    const uint32_t grid = (uint32_t)(num_segments * kStatPar), cgrid = (uint32_t)((num_segments + 255) / 256);
    hipLaunchKernelGGL((segment_chunk_sums_kernel<T, false>), dim3(grid), dim3(256), 0, st, values, counts, first_row,
                       num_segments, (const T*)nullptr, partial);
    hipLaunchKernelGGL((segment_combine_kernel<T, false>), dim3(cgrid), dim3(256), 0, st, (const T*)partial, counts, first_row,
                       num_segments, out_mean);
    hipLaunchKernelGGL((segment_chunk_sums_kernel<T, true>), dim3(grid), dim3(256), 0, st, values, counts, first_row,
                       num_segments, (const T*)out_mean, partial);
    hipLaunchKernelGGL((segment_combine_kernel<T, true>), dim3(cgrid), dim3(256), 0, st, (const T*)partial, counts, first_row,
                       num_segments, out_std);
}
