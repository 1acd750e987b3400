// vg_gram.hpp -- normal-equation build: per-image Gram matrices of the stacked row block
//   S_b = [ J_0 | J_1 | ... | J_L | r ]   (2N x W,  W = K + 6L + 1)      G_b = S_b^T S_b   (W x W)
// and their deterministic reduction over images.  J^T J, J^T r and r^T r of one residual block are the
// sub-blocks of G_b; the arrow-structured normal equations (global block + one 6x6 block per pose) are
// assembled from them (vg_solver).  This code has no counterpart in the reference tree: Ceres forms
// J^T J internally (call sites src/calibration/unified_calibration.cpp:53,426,1152).
//
// Kernels
//   vg_gram_fused_kernel      one WAVE per image: lanes evaluate corners (same device functions as the emit
//                             kernel, J never leaves the CU), rows go to a wave-private LDS tile, and
//                             v_mfma_f64_16x16x4_f64 contracts 4 rows per instruction -- the matrix core is
//                             used as the cross-lane reduction of the Gram sum.
//   vg_gram_rows_kernel       same contraction, rows read back from the materialised Ceres-layout J
//                             (the "second pass" of BASELINE.json's north_star); HBM-read bound.
//   vg_gram_reduce_kernel     fixed-order two-stage sum over images (no atomics -> run-to-run and
//                             1/2/4/8-GPU reproducible).
#pragma once

#include "vg_kernels.hpp"

namespace vg {

using f64x4 = __attribute__((ext_vector_type(4))) double;

constexpr int kGramMaxWavesPerBlock = 4;  // the host lowers it when 4 LDS tiles would not fit (wide W)
constexpr int kGramRowsPerTile = 2 * kWave;  // 64 observations x 2 rows

struct GramArgs {
    const double *frames;
    const double *board;
    const double *obs;
    const double *intr;
    // two-pass source (vg_gram_rows_kernel)
    const double *res;
    const double *jac_intr;
    const double *jac_member[kMaxChain];
    double *gram;  // [n_blocks][W*W] row-major, full symmetric
    unsigned int n_blocks;
    unsigned int N;
    int L;
    int W;
    int frame_stride_d;
    // speculative launches of the LM loop (vg_solver_impl.hpp): the kernel returns at once unless *gate == gate_expect.
    // NULL = always run.
    const int *gate;
    int gate_expect;
};

__device__ __forceinline__ bool gate_closed(const int *gate, int expect)
{
    return gate != nullptr && *reinterpret_cast<const volatile int *>(gate) != expect;
}

// D(16x16) += A(16x4) * B(4x16); lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15].
__device__ __forceinline__ f64x4 mfma_f64_16x16x4(double a, double b, f64x4 c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Store the T x T tiles of accumulators as the full symmetric W x W matrix.
// f64 C/D layout: lane l, register r holds D[row = (l>>4) + 4r][col = l&15].
template <int T, bool SKIP_LAST_DIAGONAL = false>
__device__ __forceinline__ void store_gram(const f64x4 (&acc)[T][T], double *__restrict__ g, int W, int lane)
{
    const int col = lane & 15, row0 = lane >> 4;
#pragma unroll
    for (int ti = 0; ti < T; ti++)
#pragma unroll
        for (int tj = ti; tj < T; tj++) {
            if (SKIP_LAST_DIAGONAL && ti == T - 1 && tj == T - 1) continue;  // written by the caller
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = 16 * ti + row0 + 4 * r, j = 16 * tj + col;
                if (i < W && j < W) {
                    g[i * W + j] = acc[ti][tj][r];
                    if (ti != tj) g[j * W + i] = acc[ti][tj][r];
                }
            }
        }
}

template <int T>
__device__ __forceinline__ void zero_acc(f64x4 (&acc)[T][T])
{
#pragma unroll
    for (int i = 0; i < T; i++)
#pragma unroll
        for (int j = 0; j < T; j++) acc[i][j] = f64x4{0., 0., 0., 0.};
}

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// ------------------------------------------------------------------------------------------
// fused evaluate + Gram.  One wave owns TWO consecutive images, one per 32-lane half, and walks their
// corners 32 at a time (an 8 x 12 board is 3 full steps, no idle lanes; a 64-wide step would idle a
// quarter of them).  Per step every lane evaluates one corner and writes its two rows to the wave's LDS
// tile (rows 0..63: image A, 64..127: image B); then v_mfma_f64_16x16x4_f64 contracts 4 rows per
// instruction, alternating between the A and B accumulators (two independent chains).
// dynamic LDS per wave: [128 rows][W] row tile + the two images' frames.
// T = ceil(W / 16) column tiles.
// ------------------------------------------------------------------------------------------
constexpr int kGramHalf = 32;

__host__ __device__ constexpr int gram_wave_lds_doubles(int W, int frame_stride_d)
{
    return kGramRowsPerTile * (W | 1) + 2 * frame_stride_d;  // odd LDS row stride, see the kernel
}

// CORNER (only with T == 2, W <= 20): the small (W-16) x (W-16) corner of the Gram matrix -- the 1 x 1 r^T r of Mei
// mono (W = 17), the 3 x 3 of the stereo chain (W = 19) -- would cost a whole third MFMA per 4 rows; it is
// accumulated by the lanes instead (<= 10 products per row) and reduced once per image with shuffles.
constexpr int kCornerMax = 4;

//
// RCOL (only with T == 1, W == 17 -- Mei mono: K + 6 = 16 Jacobian columns + the residual column): the Jacobian
// columns fill the 16 x 16 MFMA tile exactly, so the whole 17th row/column (J^T r and r^T r) is accumulated by the
// lanes -- 17 products per row -- instead of costing a second MFMA per 4 rows that would be 15/16 padding.
template <int MODEL, int T, bool CORNER = false, bool RCOL = false>
__global__ __launch_bounds__(kGramMaxWavesPerBlock *kWave) void vg_gram_fused_kernel(GramArgs a)
{
    static_assert(!CORNER || T == 2, "the VALU corner only exists for two column tiles");
    static_assert(!RCOL || (T == 1 && !CORNER), "the VALU residual column goes with one full column tile");
    constexpr int K = CameraTraits<MODEL>::K;
    using d2 = HIP_vector_type<double, 2>;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (gate_closed(a.gate, a.gate_expect)) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned int bA = 2 * (blockIdx.x * (blockDim.x >> 6) + wave);  // wave-uniform
    if (bA >= a.n_blocks) return;  // whole wave leaves; no workgroup barrier below
    const int W = a.W, FS = a.frame_stride_d;
    double *tile = smem + (size_t)wave * gram_wave_lds_doubles(W, FS);
    // LDS row stride: odd number of doubles.  An even W (UCM: 12, 18, 24) puts the rows of lanes l and l + 16 in the
    // same banks -- UCM mono ran at 41 us where the wider EUCM block (W = 13) took 35 us.
    const int WS = W | 1;
    double *fr_lds = tile + kGramRowsPerTile * WS;
    const int h = lane >> 5, sl = lane & (kGramHalf - 1);
    const unsigned int b = bA + h;
    const bool bvalid = b < a.n_blocks;

    // both frames are adjacent in memory: one coalesced copy into LDS
    {
        const int n_fr = (a.n_blocks - bA >= 2 ? 2 : 1) * FS;
        const double *src = a.frames + (size_t)bA * FS;
        for (int i = lane; i < n_fr; i += kWave) fr_lds[i] = src[i];
        wave_lds_fence();
    }
    const double *fr = fr_lds + (bvalid ? h : 0) * FS;

    f64x4 accA[T][T], accB[T][T];
    zero_acc<T>(accA);
    zero_acc<T>(accB);
    double cacc[kCornerMax * (kCornerMax + 1) / 2];
#pragma unroll
    for (int q = 0; q < kCornerMax * (kCornerMax + 1) / 2; q++) cacc[q] = 0.;
    const int Wc = CORNER ? W - 16 : 0;  // 1 .. kCornerMax
    double racc[RCOL ? 17 : 1];
#pragma unroll
    for (int q = 0; q < (RCOL ? 17 : 1); q++) racc[q] = 0.;
    const int c16 = lane & 15, k4 = lane >> 4;

    for (unsigned int c0 = 0; c0 < a.N; c0 += kGramHalf) {
        const unsigned int c = c0 + sl;
        const bool valid = bvalid && c < a.N;
        const unsigned int cc = c < a.N ? c : a.N - 1;
        const unsigned int bb = bvalid ? b : bA;
        const double g0 = a.board[3 * cc], g1 = a.board[3 * cc + 1], g2 = a.board[3 * cc + 2];
        const double X0 = (fr[0] * g0 + fr[1] * g1 + fr[2] * g2) + fr[9];
        const double X1 = (fr[3] * g0 + fr[4] * g1 + fr[5] * g2) + fr[10];
        const double X2 = (fr[6] * g0 + fr[7] * g1 + fr[8] * g2) + fr[11];
        const d2 ob = reinterpret_cast<const d2 *>(a.obs)[(size_t)bb * a.N + cc];
        CornerEval<K> e;
        eval_corner_fast<MODEL>(a.intr, X0, X1, X2, e);

        // rows are written unconditionally; lanes without a corner then overwrite theirs with zeros (a branch
        // no lane takes on full boards) -- cheaper than a select per element
        double *ru = tile + (size_t)(kWave * h + 2 * sl) * WS, *rv = ru + WS;
#pragma unroll
        for (int i = 0; i < K; i++) {
            ru[i] = e.Ju[i];
            rv[i] = e.Jv[i];
        }
        for (int l = 0; l < a.L; l++) {
            double rows[12];
            pose_rows_fast(e.P, X0, X1, X2, fr + 12 + 21 * l, rows);
#pragma unroll
            for (int j = 0; j < 6; j++) {
                ru[K + 6 * l + j] = rows[j];
                rv[K + 6 * l + j] = rows[6 + j];
            }
        }
        // residual column; a failed projection contributes the in-band 1e15 exactly as it would
        // inside Ceres (calib_cost_functions.cpp:66-70)
        ru[W - 1] = e.ok ? e.u - ob.x : kDoubleBig;
        rv[W - 1] = e.ok ? e.v - ob.y : kDoubleBig;
        if (!valid) {
            for (int i = 0; i < W; i++) {
                ru[i] = 0.;
                rv[i] = 0.;
            }
        }
        if (CORNER) {
            // this lane's own two rows, columns 16 .. W-1 (just written; DS ops of a wave execute in order)
            double cu[kCornerMax], cv[kCornerMax];
#pragma unroll
            for (int q = 0; q < kCornerMax; q++) {
                cu[q] = q < Wc ? ru[16 + q] : 0.;
                cv[q] = q < Wc ? rv[16 + q] : 0.;
            }
#pragma unroll
            for (int r = 0, q = 0; r < kCornerMax; r++)
#pragma unroll
                for (int c = r; c < kCornerMax; c++, q++) cacc[q] += cu[r] * cu[c] + cv[r] * cv[c];
        }
        if constexpr (RCOL) {
            // this lane's own two rows (just written, zeroed when the lane has no corner): column 16 against all 17
            const double su = ru[16], sv = rv[16];
#pragma unroll
            for (int q = 0; q < 16; q++) racc[q] += ru[q] * su + rv[q] * sv;
            racc[16] += su * su + sv * sv;
        }
        wave_lds_fence();

        // always 16 groups of 4 rows per image: rows of lanes without a corner are zero, so a ragged last
        // step only wastes matrix-pipe time, and the fixed trip count lets the LDS reads be pipelined
        constexpr int n_steps = kGramHalf / 2;
        const double *rowA = tile + (size_t)k4 * WS, *rowB = rowA + (size_t)kWave * WS;
#pragma unroll 4
        for (int t = 0; t < n_steps; t++) {
            double vA[T], vB[T];
#pragma unroll
            for (int j = 0; j < T; j++) {
                const int col = 16 * j + c16;
                vA[j] = col < W ? rowA[(size_t)(4 * t) * WS + col] : 0.;
                vB[j] = col < W ? rowB[(size_t)(4 * t) * WS + col] : 0.;
            }
#pragma unroll
            for (int ti = 0; ti < T; ti++)
#pragma unroll
                for (int tj = ti; tj < T; tj++) {
                    if (CORNER && ti == 1) continue;  // the corner tile is accumulated by the lanes
                    accA[ti][tj] = mfma_f64_16x16x4(vA[ti], vA[tj], accA[ti][tj]);
                    accB[ti][tj] = mfma_f64_16x16x4(vB[ti], vB[tj], accB[ti][tj]);
                }
        }
        wave_lds_fence();
    }
    store_gram<T, CORNER>(accA, a.gram + (size_t)bA * W * W, W, lane);
    if (bA + 1 < a.n_blocks) store_gram<T, CORNER>(accB, a.gram + (size_t)(bA + 1) * W * W, W, lane);
    if constexpr (RCOL) {
        // sum over the 32 lanes of each image (fixed butterfly inside the half-wave), lane 0 of the half stores
#pragma unroll
        for (int q = 0; q < 17; q++)
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) racc[q] += __shfl_xor(racc[q], off, kWave);
        if (sl == 0 && bvalid) {
            double *g = a.gram + (size_t)b * W * W;
#pragma unroll
            for (int q = 0; q < 16; q++) {
                g[q * W + 16] = racc[q];
                g[16 * W + q] = racc[q];
            }
            g[16 * W + 16] = racc[16];
        }
    }
    if (CORNER) {
        // sum over the 32 lanes of each image (fixed butterfly inside the half-wave), lane 0 of the half stores
#pragma unroll
        for (int q = 0; q < kCornerMax * (kCornerMax + 1) / 2; q++)
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) cacc[q] += __shfl_xor(cacc[q], off, kWave);
        if (sl == 0 && bvalid) {
            double *g = a.gram + (size_t)b * W * W;
#pragma unroll
            for (int r = 0, q = 0; r < kCornerMax; r++)
#pragma unroll
                for (int c = r; c < kCornerMax; c++, q++)
                    if (c < Wc) {
                        g[(16 + r) * W + 16 + c] = cacc[q];
                        g[(16 + c) * W + 16 + r] = cacc[q];
                    }
        }
    }
}

// ------------------------------------------------------------------------------------------
// two-pass: Gram of the materialised rows (the "second pass" over J).  Same work split and the same
// contraction order as the fused kernel -- one wave per pair of images, A / B accumulators alternating,
// 4 rows per MFMA in increasing row order -- so both paths give bit-identical matrices.  Every MFMA
// operand is gathered straight from the Ceres-layout arrays (per 4 rows: 4K, 24 and 4 consecutive
// doubles); kRowsUnroll groups are loaded ahead of their MFMAs to keep loads in flight.
// ------------------------------------------------------------------------------------------
constexpr int kRowsUnroll = 8;

template <int T>
__global__ __launch_bounds__(kGramMaxWavesPerBlock *kWave) void vg_gram_rows_kernel(GramArgs a, int K)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned int bA = 2 * (blockIdx.x * (blockDim.x >> 6) + wave);
    if (bA >= a.n_blocks) return;
    const bool hasB = bA + 1 < a.n_blocks;
    const int W = a.W;
    const int c = lane & 15, k = lane >> 4;
    const unsigned int rows = 2 * a.N;

    // per column tile: which array this lane reads, its row stride and offset inside the row
    const double *src[T];
    int stride[T];
    size_t img_stride[T];  // distance between image A's and image B's rows in that array
#pragma unroll
    for (int j = 0; j < T; j++) {
        const int col = 16 * j + c;
        src[j] = nullptr;
        stride[j] = 0;
        img_stride[j] = 0;
        if (col < K) {
            src[j] = a.jac_intr + (size_t)bA * rows * K + col;
            stride[j] = K;
        } else if (col < W - 1) {
            const int l = (col - K) / 6;
            src[j] = a.jac_member[l] + (size_t)bA * rows * 6 + (col - K - 6 * l);
            stride[j] = 6;
        } else if (col == W - 1) {
            src[j] = a.res + (size_t)bA * rows;
            stride[j] = 1;
        }
        img_stride[j] = (size_t)rows * stride[j];
    }
    f64x4 accA[T][T], accB[T][T];
    zero_acc<T>(accA);
    zero_acc<T>(accB);
    // the fused kernel contracts each image in steps of 32 corners = 16 groups of 4 rows, zero-padded at the
    // end of the image; reproduce exactly that sequence of (row group -> MFMA) so the sums round identically
    const unsigned int n_chunks = (a.N + kGramHalf - 1) / kGramHalf;
    for (unsigned int ch = 0; ch < n_chunks; ch++) {
        const unsigned int row0 = ch * 2 * kGramHalf;
        for (int t0 = 0; t0 < kGramHalf / 2; t0 += kRowsUnroll) {
            double vA[kRowsUnroll][T], vB[kRowsUnroll][T];
#pragma unroll
            for (int u = 0; u < kRowsUnroll; u++) {
                const unsigned int row = row0 + 4 * (t0 + u) + k;
#pragma unroll
                for (int j = 0; j < T; j++) {
                    const bool ok = src[j] && row < rows;
                    vA[u][j] = ok ? src[j][(size_t)row * stride[j]] : 0.;
                    vB[u][j] = (ok && hasB) ? src[j][img_stride[j] + (size_t)row * stride[j]] : 0.;
                }
            }
#pragma unroll
            for (int u = 0; u < kRowsUnroll; u++)
#pragma unroll
                for (int ti = 0; ti < T; ti++)
#pragma unroll
                    for (int tj = ti; tj < T; tj++) {
                        accA[ti][tj] = mfma_f64_16x16x4(vA[u][ti], vA[u][tj], accA[ti][tj]);
                        accB[ti][tj] = mfma_f64_16x16x4(vB[u][ti], vB[u][tj], accB[ti][tj]);
                    }
        }
    }
    store_gram<T>(accA, a.gram + (size_t)bA * W * W, W, lane);
    if (hasB) store_gram<T>(accB, a.gram + (size_t)(bA + 1) * W * W, W, lane);
}

// ------------------------------------------------------------------------------------------
// Deterministic sum over images, two launches, no atomics (run-to-run and 1/2/4/8-GPU reproducible):
//   stage 1  vg_gram_slab_sum_kernel   workgroup p adds kSlab consecutive images, entry-parallel; the kSlab loads
//                                      of a lane are independent (all in flight), the adds a fixed tree
//   stage 2  vg_gram_final_sum_kernel  one workgroup per 4 entries: 64 lanes stride over the partials in
//                                      a fixed order, then a fixed-order wave reduction
// ------------------------------------------------------------------------------------------
constexpr int kSlab = 32;

__device__ __forceinline__ void gram_slab_sum_body(const double *__restrict__ in, unsigned int n_items, int entries,
                                                   double *__restrict__ out, unsigned int slab)
{
    const unsigned int i0 = slab * kSlab;
    for (int e = threadIdx.x; e < entries; e += blockDim.x) {
        double v[kSlab];
#pragma unroll
        for (int k = 0; k < kSlab; k++) v[k] = (i0 + k < n_items) ? in[(size_t)(i0 + k) * entries + e] : 0.;
        // fixed pairwise tree
#pragma unroll
        for (int w = 1; w < kSlab; w *= 2)
#pragma unroll
            for (int k = 0; k + w < kSlab; k += 2 * w) v[k] += v[k + w];
        out[(size_t)slab * entries + e] = v[0];
    }
}

#ifdef VG_TU_GRAM  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(256) void vg_gram_slab_sum_kernel(const double *__restrict__ in, unsigned int n_items,
                                                                int entries, double *__restrict__ out)
{
    gram_slab_sum_body(in, n_items, entries, out, blockIdx.x);
}
#endif

// "The launch is over" for a SPINNING host (the host-driven LM loop reads what the sum kernels stored in pinned memory): every
// workgroup makes its stores visible system-wide and counts itself; the last one stores the sequence number the host waits
// for (system-scope release) and re-arms the counter.  hipStreamSynchronize costs ~3.5 us more per round trip
// (tools/exp/host_wait.hip); host_seq = NULL: nothing happens.
struct HostSignal {
    unsigned int *counter = nullptr;         // device word, zero between launches
    unsigned long long *host_seq = nullptr;  // pinned word
    unsigned long long seq = 0;
};

__device__ __forceinline__ void signal_host_when_last(const HostSignal &h)
{
    if (!h.host_seq) return;
    // The results went to fine-grained host memory (uncached on the device: a store is on its way to the host once it is
    // acknowledged): waiting for this thread's stores is all a workgroup owes before it counts itself.  A system-scope fence
    // here writes the L2 back in EVERY workgroup -- the rig's 800-workgroup partial-sum launch took 50 us longer with it.
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int done = __hip_atomic_fetch_add(h.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
            __hip_atomic_store(h.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(h.host_seq, h.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// the same report from a launch of its own behind a kernel with too many workgroups to count (one atomic per workgroup of the
// rig's 800-workgroup partial-sum launch cost more than the runtime's wait saves)
#ifdef VG_TU_SOLVER
__global__ void vg_host_flag_kernel(HostSignal h)
{
    __hip_atomic_store(h.host_seq, h.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif

__device__ __forceinline__ void gram_final_sum_body(const double *__restrict__ in, unsigned int n_items, int entries,
                                                    double *__restrict__ out, unsigned int block)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int e = block * (blockDim.x >> 6) + (threadIdx.x >> 6);  // one wave per entry
    if (e >= entries) return;
    // four independent partial sums per lane (item i goes to accumulator (i / 64) % 4): the loads of a group of four
    // are in flight together -- with thousands of partials a single dependent chain made this kernel latency bound
    double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
    unsigned int i = lane;
    for (; i + 3 * kWave < n_items; i += 4 * kWave) {
        const double v0 = in[(size_t)i * entries + e], v1 = in[(size_t)(i + kWave) * entries + e];
        const double v2 = in[(size_t)(i + 2 * kWave) * entries + e], v3 = in[(size_t)(i + 3 * kWave) * entries + e];
        s0 += v0;
        s1 += v1;
        s2 += v2;
        s3 += v3;
    }
    for (; i < n_items; i += kWave) s0 += in[(size_t)i * entries + e];
    double s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, kWave);  // fixed butterfly order
    if (lane == 0) out[e] = s;
}

#ifdef VG_TU_GRAM  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(256) void vg_gram_final_sum_kernel(const double *__restrict__ in, unsigned int n_items,
                                                                 int entries, double *__restrict__ out)
{
    gram_final_sum_body(in, n_items, entries, out, blockIdx.x);
}
#endif

// The five scalar sums of an LM step (per-workgroup partials of the back-substitution -> out[5], the final-sum order) and,
// with them, the step's max |g_pose| (a bit pattern kept by atomicMax) copied to `gmax_out`: the host-driven loop points both
// outputs at pinned host memory -- a store at the end of a kernel instead of a copy command behind it.
#ifdef VG_TU_SOLVER  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(256) void vg_step_scalars_kernel(const double *__restrict__ in, unsigned int n_items, double *__restrict__ out,
                                                               const unsigned long long *__restrict__ gmax_bits, unsigned long long *gmax_out)
{
    gram_final_sum_body(in, n_items, 5, out, blockIdx.x);
    if (blockIdx.x == 1 && threadIdx.x == 0 && gmax_out) *gmax_out = *gmax_bits;
}
#endif

// The same two stages for SEVERAL datasets in one launch each (a rig has one Gram array per camera; their sums are
// launch-latency bound, so four datasets cost two launches instead of eight).  Identical arithmetic and order per
// dataset as the single-dataset kernels.
struct SumDataset {
    const double *gram;     // [n_items][entries]
    double *partials;       // [n_slabs][entries]
    double *out;            // [entries]
    unsigned int n_items, n_slabs;
    int entries;
    unsigned int first_slab_block;   // first workgroup of this dataset in the slab launch
    unsigned int first_final_block;  // ... in the final launch (4 entries per workgroup)
};

#ifdef VG_TU_SOLVER  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(256) void vg_gram_slab_sum_multi_kernel(const SumDataset *__restrict__ ds, int n_ds)
{
    int d = 0;
    while (d + 1 < n_ds && blockIdx.x >= ds[d + 1].first_slab_block) d++;
    const SumDataset D = ds[d];
    gram_slab_sum_body(D.gram, D.n_items, D.entries, D.partials, blockIdx.x - D.first_slab_block);
}
#endif

#ifdef VG_TU_SOLVER  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(256) void vg_gram_final_sum_multi_kernel(const SumDataset *__restrict__ ds, int n_ds)
{
    int d = 0;
    while (d + 1 < n_ds && blockIdx.x >= ds[d + 1].first_final_block) d++;
    const SumDataset D = ds[d];
    gram_final_sum_body(D.partials, D.n_slabs, D.entries, D.out, blockIdx.x - D.first_final_block);
}
#endif

}  // namespace vg
