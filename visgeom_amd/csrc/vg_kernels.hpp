// vg_kernels.hpp -- HIP kernels of the residual / Jacobian hot path (gfx950, wave64).
//
//   kernel 1  vg_chain_prep_multi_kernel   one lane per residual block (image) of every dataset: transform chain -> frame
//   kernel 2  vg_emit_kernel         one lane per (image, corner) observation: residual pair +
//                                    the 2 x (K + 6L) Jacobian rows, written in the Ceres block layout
//
// Both are HBM-streaming kernels (about 1 flop per byte); nothing here is GEMM shaped.  Design notes
// and the roofline arithmetic are in DESIGN.md sections 4-5.
#pragma once

#include "vg_camera.hpp"

namespace vg {

constexpr int kMaxChain = 5;
constexpr int kEmitThreads = 256;  // 4 waves
constexpr int kWave = 64;
constexpr double kDoubleBig = 1e15;  // include/std.h:71

// where chain member l of block b lives:  params + base[l] + stride[l] * seq_index[b]
// (stride 0 = global transform, 6 = sequence transform; unified_calibration.h:161-165)
struct ChainDesc {
    int L;
    int status[kMaxChain];
    long long base[kMaxChain];
    long long stride[kMaxChain];
};

// ------------------------------------------------------------------------------------------
// kernel 1: chain prep (vg_chain_prep_multi_kernel below; a table variant for problems of more than kPrepMax datasets).  64-thread
// workgroups so that 10 k images spread over ~157 CUs with one wave each: the small launches are latency bound (a dependent
// chain of sqrt / sincos / divisions), the large ones (>= 512 waves) bandwidth bound on their frame stores.
// ------------------------------------------------------------------------------------------
// all datasets of a problem in ONE launch: the kernel is latency bound, so four datasets cost the same ~7 us as one
struct PrepDataset {
    ChainDesc chain;
    const int *seq_index;
    double *frames;
    long long first;   // global index of the dataset's first block (host bookkeeping)
    long long count;
    int frame_stride_d;
};

// Up to kPrepMax datasets travel BY VALUE in the kernel arguments and every wave belongs to ONE dataset (each dataset's lanes
// are padded to whole waves): the wave finds its dataset with scalar compares and reads the descriptor -- chain length,
// directions, bases, strides, frame pointer -- with scalar loads from the kernel-argument segment.  Round 4's kernel read a
// descriptor TABLE in global memory per lane: `first` of the next dataset -> the descriptor's pointers -> seq_index[b] ->
// base / stride -> the parameters were five dependent memory round trips in front of a two-microsecond walk (8.4-11.9 us per
// launch, SQ "wait any" 0.56); now the parameters are the first vector load.
constexpr int kPrepMax = 8;
struct PrepMultiArgs {
    PrepDataset ds[kPrepMax];
    unsigned int first_wave[kPrepMax + 1];  // first wave (= 64-thread workgroup) of each dataset in this launch
    int n;
    int staged;   // 1: the frames leave through LDS as linear 16-byte stores (launches of >= kPrepStagedMinWaves waves: bandwidth matters)
};
constexpr unsigned int kPrepStagedMinWaves = 512;   // 32 768 images

#ifdef VG_TU_CORE  // this kernel is launched by one translation unit only; the others see the header without it
// problems of more than kPrepMax datasets: the descriptors in a table in global memory, one lane per block across all datasets
__global__ __launch_bounds__(64) void vg_chain_prep_table_kernel(const double *__restrict__ params,
                                                                  const PrepDataset *__restrict__ dsets, int n_dsets,
                                                                  long long total_blocks)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total_blocks) return;
    int d = 0;
    while (d + 1 < n_dsets && t >= dsets[d + 1].first) d++;
    const PrepDataset *D = dsets + d;
    const long long b = t - D->first;
    const int *seq = D->seq_index;
    const long long si = seq ? (long long)seq[b] : b;
    const long long *base = D->chain.base, *stride = D->chain.stride;
    build_frame(D->chain.L, D->chain.status, [&](int l) { return params + base[l] + stride[l] * si; },
                D->frames + b * D->frame_stride_d);
}

// All members' parameters are requested in one round in front of the walk.  Large launches (m.staged): the wave's 64 frames are
// staged in LDS (dynamic: 64 x the launch's widest frame) and leave as one linear run of 16-byte stores -- written by the lanes
// themselves, every lane its own 264-byte frame, the kernel ran at 1.3 TB/s and cost 20 us per 100 k images, 214 us at 1 M
// (now 13 / 67 us; profiles/r06_emit_drop.md).
__global__ __launch_bounds__(64) void vg_chain_prep_multi_kernel(const double *__restrict__ params, PrepMultiArgs m)
{
    using d2 = HIP_vector_type<double, 2>;
    extern __shared__ __attribute__((aligned(16))) double prep_tile[];
    int d = 0;
    while (d + 1 < m.n && blockIdx.x >= m.first_wave[d + 1]) d++;
    const PrepDataset &D = m.ds[d];
    const long long b0 = (long long)(blockIdx.x - m.first_wave[d]) * 64;
    const long long b = b0 + threadIdx.x;
    const int stride = D.frame_stride_d, L = D.chain.L;
    if (b < D.count) {
        const int *seq = D.seq_index;
        const long long si = seq ? (long long)seq[b] : b;
        double xi[kMaxChain][6];
#pragma unroll
        for (int l = 0; l < kMaxChain; l++)
            if (l < L) {
                const double *src = params + D.chain.base[l] + D.chain.stride[l] * si;
#pragma unroll
                for (int k = 0; k < 6; k++) xi[l][k] = src[k];
            }
        auto walk = [&](double *frame) {   // build_frame(), its member loop unrolled over the preloaded members
            ChainState s;
            chain_state_init(s);
#pragma unroll
            for (int l = 0; l < kMaxChain; l++)
                if (l < L) {
                    const Quat q1 = chain_acc_quat(s);
                    chain_walk_member(s, q1, xi[l], D.chain.status[l] != 0, frame + 12 + 21 * l);
                }
            chain_finish(s, frame);
        };
        // small launches (a stereo pair's 64 waves, a rig's 316) are pure latency: their lanes store their frames themselves, as
        // before round 6 -- the LDS round trip would only add to the dependent chain (7.0 -> 7.6 us for the stereo pair)
        if (!m.staged) {
            walk(D.frames + b * stride);
            return;
        }
        walk(prep_tile + threadIdx.x * stride);
    }
    if (!m.staged) return;
    // one wave per workgroup: LDS executes its DS operations in order, only the compiler needs the fence
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const long long left = D.count - b0;
    const int total = (int)(left < 64 ? left : 64) * stride;   // doubles of this wave's frames, consecutive in memory
    double *dst = D.frames + b0 * stride;
    if ((reinterpret_cast<unsigned long long>(dst) & 15ull) == 0) {
        for (int i = threadIdx.x; i < (total >> 1); i += 64) reinterpret_cast<d2 *>(dst)[i] = reinterpret_cast<const d2 *>(prep_tile)[i];
        if ((total & 1) && threadIdx.x == 0) dst[total - 1] = prep_tile[total - 1];
    } else {
        for (int i = threadIdx.x; i < total; i += 64) dst[i] = prep_tile[i];
    }
}
#endif

// ------------------------------------------------------------------------------------------
// kernel 2: emit.
// ------------------------------------------------------------------------------------------
struct EmitArgs {
    const double *frames;  // [n_blocks][frame_stride]
    const double *board;   // [N][3]
    const double *obs;     // [n_blocks][N][2]
    const double *intr;    // [K]
    double *res;           // [n_blocks][2N]
    double *jac_intr;      // [n_blocks][2N][K] or NULL
    double *jac_member[kMaxChain];  // [n_blocks][2N][6] or NULL
    unsigned long long *failed;     // failed-projection counter word: (epoch << 40) | count  (may be NULL)
    unsigned long long epoch;       // evaluation number (24 bits); a stale epoch in the word means count 0
    // INLINE_CHAIN variant only (single-member DIRECT chain): where the member's 6-vector of image b of this launch
    // lives: chain_params + chain_stride * (seq_index ? seq_index[b] : first_block + b)
    const double *chain_params;
    const int *seq_index;
    long long chain_stride;
    long long first_block;
    unsigned int n_obs;    // n_blocks * N  (< 2^31 per launch; the host chunks larger problems)
    unsigned int N;
    int L;
    int frame_stride_d;
    int nt_stores;         // 1: the launch's output streams past the Infinity Cache -> non-temporal stores (stream_store16)
    unsigned int map_window;  // tile map of the launch: 0 = every XCD one contiguous eighth of the tiles; W > 0 = windows of 8 W tiles,
                              // XCD x the x-th run of W tiles in each window (W = 1: the linear map); xcd_window_block, kEmitMapWindow
};

// Each lane holds the 2S doubles of its observation's two rows; the wave's 64 observations are one
// contiguous 1024*S-byte run of the output.  Lanes write their rows to the wave's private LDS tile
// (16 B stores at a 16*S-byte lane stride) and the tile is then streamed out linearly, 16 B per lane
// per store -> every global store instruction writes 1 KiB of consecutive bytes.
// The 16-byte store of the output stream, plain or with the non-temporal hint (`global_store_dwordx4 ... nt`), chosen per LAUNCH by
// the host (EmitArgs::nt_stores, a wave-uniform branch).  A launch whose output fits the 256 MiB Infinity Cache keeps plain
// stores: the consumer (second-pass Gram, a copy engine, the next evaluation overwriting the same rows) finds the lines there.
// A launch that streams past the cache sets the hint: a line kept behind the write only displaces the next lines of the same
// stream (same box, alternating builds, profiles/r05c_emit_sweep_ab.txt: EUCM 12.5 k images = 250 MB 57 -> 45 us, 25 k 105 -> 83,
// 50 k = 1 GB 201 -> 159 us, 0.67 -> 0.84 of the HBM peak; inside the cache the hint costs: 10 k images 36 -> 41 us).
__device__ __forceinline__ void stream_store16(HIP_vector_type<double, 2> *dst, const HIP_vector_type<double, 2> &v, bool nt)
{
    if (nt) {
        __builtin_nontemporal_store(v.x, &dst->x);
        __builtin_nontemporal_store(v.y, &dst->y);
    } else {
        *dst = v;
    }
}

template <int S>
__device__ __forceinline__ void wave_store_rows(double *__restrict__ stage, const double *vals,
                                                double *__restrict__ out_tile, int n_valid_obs, int lane, bool nt = false)
{
    using d2 = HIP_vector_type<double, 2>;
    d2 *st = reinterpret_cast<d2 *>(stage);
#pragma unroll
    for (int i = 0; i < S; i++) {
        d2 v;
        v.x = vals[2 * i];
        v.y = vals[2 * i + 1];
        st[lane * S + i] = v;
    }
    // wave-private tile: LDS executes one wave's DS ops in order, only the compiler needs fencing
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int n16 = n_valid_obs * S;
    d2 *dst = reinterpret_cast<d2 *>(out_tile);
#pragma unroll
    for (int k = 0; k < S; k++) {
        const int idx = k * kWave + lane;
        if (idx < n16) stream_store16(dst + idx, st[idx], nt);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Same tile for every model: rows wider than a pose block (Mei, K = 10) are staged in two half-waves of 32
// observations each, so the LDS footprint -- and with it the number of resident waves that keep stores in flight --
// does not depend on K (with a 10 KiB tile per wave only 3 workgroups fit a CU and Mei streamed at 5.2 TB/s).
constexpr int kStageRowDoubles = 6;

template <int MODEL>
constexpr int emit_stage_doubles_per_wave()
{
    return 2 * kWave * kStageRowDoubles;
}

// wave_store_rows for S > kStageRowDoubles: lanes [32h, 32h + 32) stage their rows, the whole wave streams the
// 32 * 16 * S contiguous bytes out, h = 0, 1.
template <int S>
__device__ __forceinline__ void wave_store_rows_halves(double *__restrict__ stage, const double *vals,
                                                       double *__restrict__ out_tile, int n_valid_obs, int lane, bool nt = false)
{
    static_assert(S <= 2 * kStageRowDoubles, "half a wave of rows must fit the tile");
    using d2 = HIP_vector_type<double, 2>;
    d2 *st = reinterpret_cast<d2 *>(stage);
    constexpr int kHalf = kWave / 2;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if ((lane >> 5) == h) {
#pragma unroll
            for (int i = 0; i < S; i++) {
                d2 v;
                v.x = vals[2 * i];
                v.y = vals[2 * i + 1];
                st[(lane & (kHalf - 1)) * S + i] = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int nv = n_valid_obs - h * kHalf;
        nv = nv < 0 ? 0 : (nv > kHalf ? kHalf : nv);
        const int n16 = nv * S;
        d2 *dst = reinterpret_cast<d2 *>(out_tile) + h * kHalf * S;
#pragma unroll
        for (int k = 0; k < (kHalf * S + kWave - 1) / kWave; k++) {
            const int idx = k * kWave + lane;
            if (idx < n16) stream_store16(dst + idx, st[idx], nt);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// Workgroup -> tile mapping for the streaming kernels.  Block b runs on XCD b % 8 (observed dispatch order, used for
// speed only -- any mapping is correct).  Handing XCD x the x-th contiguous eighth of the output instead of every
// eighth tile keeps each die's write stream in its own address range: the same 2.15 GB streaming write runs at
// 6.63 TB/s instead of 6.02 TB/s (tools/exp/store_sweep.hip); below the 256 MiB Infinity Cache it makes no difference.
__device__ __forceinline__ unsigned int xcd_contiguous_block(unsigned int b, unsigned int n)
{
    constexpr unsigned int kXcds = 8;
    const unsigned int x = b % kXcds, j = b / kXcds, q = n / kXcds, r = n % kXcds;
    return x * q + (x < r ? x : r) + j;  // bijective on [0, n): XCD x owns q (+1 for x < r) consecutive tiles
}

// The same with the dies advancing TOGETHER through the output: the tiles are cut into windows of 8 W tiles and XCD x owns the
// x-th run of W consecutive tiles of every window; the last, partial window is split into contiguous eighths like the whole
// range above.  W >= n / 8 is xcd_contiguous_block, W = 1 the linear map.  Bijective on [0, n).
// Why (profiles/r06_emit_drop.md): with one contiguous eighth per die a 2 GB evaluation keeps 8 x 3 write cursors >= 110 MB
// apart, and on most boxes / placements that costs a tenth of the rate (EUCM 100 k images 395 us against 362 us with W = 16
// on the same box and arrays; where the eighths run at full rate, 341 us, the windows give 344 us); inside the Infinity Cache
// the maps do not differ.  The counters say it is the DRAM side: no address-translation misses (TCP_UTCL1_TRANSLATION_MISS
// 1e3 of 4e7 requests), FEWER L2 -> fabric credit stalls and fewer writes in flight than at 1 GB, i.e. requests retire slower.
constexpr unsigned int kEmitMapWindow = 16;   // runs of 16 tiles: 384 KiB of a 6-column Jacobian array per die and window (launches >= 1.2 GB: emit_map_window, vg_capi.hip)
__device__ __forceinline__ unsigned int xcd_window_block(unsigned int b, unsigned int n, unsigned int W)
{
    const unsigned int x = b & 7u, j = b >> 3, w = j / W, i = j - w * W, base = w * 8u * W, rem = n - base;
    if (rem >= 8u * W) return base + x * W + i;
    const unsigned int q = rem >> 3, r = rem & 7u;
    return base + x * q + (x < r ? x : r) + i;
}

// dynamic LDS: 4 wave tiles, then (FRAMES_LDS) the frames of the images this workgroup touches
// INLINE_CHAIN (with FRAMES_LDS, chain = one DIRECT member): the workgroup derives the <= 4 frames it needs itself
// (thread f walks image b_first + f with build_frame_single_direct) instead of reading them from the chain-prep
// kernel's output -- a full evaluation is then ONE launch.
// One 256-observation tile of one dataset (o0 = first observation of the tile).
template <int MODEL, bool WANT_JAC, bool FRAMES_LDS, bool INLINE_CHAIN>
__device__ __forceinline__ void emit_tile(const EmitArgs &a, const unsigned int o0)
{
    static_assert(!INLINE_CHAIN || FRAMES_LDS, "the inline chain writes its frames to LDS");
    constexpr int K = CameraTraits<MODEL>::K;
    using d2 = HIP_vector_type<double, 2>;
    extern __shared__ __attribute__((aligned(16))) double smem[];

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    const unsigned int o = o0 + tid;
    const bool active = o < a.n_obs;
    const unsigned int oc = active ? o : a.n_obs - 1;
    const unsigned int b = oc / a.N;
    const unsigned int c = oc - b * a.N;

    // issued before the frames are staged / derived: their latency overlaps the barrier below
    const double g0 = a.board[3 * c], g1 = a.board[3 * c + 1], g2 = a.board[3 * c + 2];
    const d2 ob = reinterpret_cast<const d2 *>(a.obs)[oc];

    const double *fr;
    if (FRAMES_LDS) {
        double *fr_lds = smem + (kEmitThreads / kWave) * emit_stage_doubles_per_wave<MODEL>();
        const unsigned int b_first = o0 / a.N;
        const unsigned int o_last = (o0 + kEmitThreads - 1 < a.n_obs) ? o0 + kEmitThreads - 1 : a.n_obs - 1;
        const unsigned int nf = o_last / a.N - b_first + 1;
        if (INLINE_CHAIN) {
            // a tile of 256 consecutive observations touches at most 256 images: at most one frame per thread (as a loop
            // the walk was scheduled across iterations and cost 16 more VGPRs: 114 instead of 98)
            if (const unsigned int f = tid; f < nf) {
                const long long bi = (long long)b_first + f;
                const long long si = a.seq_index ? (long long)a.seq_index[bi] : a.first_block + bi;
#ifdef VG_EMIT_REFERENCE_WALK   // A/B library: the reference-order walk of rounds 1-5 (~300 dependent instructions)
                build_frame_single_direct(a.chain_params + a.chain_stride * si, fr_lds + f * a.frame_stride_d);
#else
                {   // The short walk of the Gram kernels (one sincos at the half angle, R12 = I, M12 from uhat^2 = u u^T - I: the
                    // reference's frame to 1e-16; inside and just above its first-order branches it IS the reference-order
                    // routine).  While <= 4 lanes walk, the tile's other 252 wait at the barrier below with no store in
                    // flight: a third of the chain gone is 36.3 -> 34.1 us at 10 k images, 22.0 -> 19.8 at 5 k, 161 -> 152 at 50 k
                    // (same box, alternating; profiles/r06q_emit_fastwalk_ab.txt).  The corner arithmetic stays in reference order.
                    double xi_r[6];
#pragma unroll
                    for (int k = 0; k < 6; k++) xi_r[k] = (a.chain_params + a.chain_stride * si)[k];
                    build_frame_single_direct_fast(xi_r, fr_lds + f * a.frame_stride_d);
                }
#endif
            }
        } else {
            const int n16 = (int)(nf * (unsigned)a.frame_stride_d) >> 1;
            const d2 *src = reinterpret_cast<const d2 *>(a.frames + (size_t)b_first * a.frame_stride_d);
            d2 *dst = reinterpret_cast<d2 *>(fr_lds);
            for (int i = tid; i < n16; i += kEmitThreads) dst[i] = src[i];
        }
        __syncthreads();
        fr = fr_lds + (b - b_first) * a.frame_stride_d;
    } else {
        fr = a.frames + (size_t)b * a.frame_stride_d;
    }

    // pointCam = R(xiAcc.rot) * grid + xiAcc.trans     calib_cost_functions.cpp:49-50
    const double X0 = (fr[0] * g0 + fr[1] * g1 + fr[2] * g2) + fr[9];
    const double X1 = (fr[3] * g0 + fr[4] * g1 + fr[5] * g2) + fr[10];
    const double X2 = (fr[6] * g0 + fr[7] * g1 + fr[8] * g2) + fr[11];

    CornerEval<K> e;
    eval_corner<MODEL, WANT_JAC, WANT_JAC>(a.intr, X0, X1, X2, e);

    // residual pair, or the in-band failure value       calib_cost_functions.cpp:57-71
    d2 r;
    r.x = e.ok ? e.u - ob.x : kDoubleBig;
    r.y = e.ok ? e.v - ob.y : kDoubleBig;
    if (active) stream_store16(reinterpret_cast<d2 *>(a.res) + o, r, a.nt_stores != 0);

    if (a.failed) {
        // Failures are rare: the counter is never zeroed (an 8-byte hipMemsetAsync is a whole 5 us fill
        // kernel).  The word carries the evaluation epoch; the first failing wave of an evaluation
        // replaces a stale word, later ones add to it.
        const unsigned long long m = __ballot(active && !e.ok);
        if (m && lane == 0) {
            const unsigned long long n = (unsigned long long)__popcll(m);
            unsigned long long old = *a.failed, assumed;
            do {
                assumed = old;
                const unsigned long long cnt = (assumed >> 40) == a.epoch ? (assumed & ((1ull << 40) - 1)) + n : n;
                old = atomicCAS(a.failed, assumed, (a.epoch << 40) | cnt);
            } while (old != assumed);
        }
    }

    if (WANT_JAC) {
        double *stage = smem + wave * emit_stage_doubles_per_wave<MODEL>();
        const unsigned int ow = o0 + wave * kWave;  // first observation of this wave
        int n_valid = 0;
        if (ow < a.n_obs) n_valid = (a.n_obs - ow < (unsigned)kWave) ? (int)(a.n_obs - ow) : kWave;

        // intrinsic block, rows 2i / 2i+1 of [2N x K]       calib_cost_functions.cpp:105-114
        if (a.jac_intr) {
            double rows[2 * K];
#pragma unroll
            for (int i = 0; i < K; i++) {
                rows[i] = e.Ju[i];
                rows[K + i] = e.Jv[i];
            }
            if (K > kStageRowDoubles) wave_store_rows_halves<K>(stage, rows, a.jac_intr + (size_t)ow * (2 * K), n_valid, lane, a.nt_stores != 0);
            else wave_store_rows<(K > kStageRowDoubles ? 1 : K)>(stage, rows, a.jac_intr + (size_t)ow * (2 * K), n_valid, lane, a.nt_stores != 0);
        }
        // pose blocks, u-row at +12i, v-row at +12i+6       calib_cost_functions.cpp:93-101
        for (int l = 0; l < a.L; l++) {
            double *Jm = a.jac_member[l];
            if (!Jm) continue;
            double rows[12];
            pose_rows(e.P, X0, X1, X2, fr + 12 + 21 * l, rows);
            wave_store_rows<6>(stage, rows, Jm + (size_t)ow * 12, n_valid, lane, a.nt_stores != 0);
        }
    }
}

template <int MODEL, bool WANT_JAC, bool FRAMES_LDS, bool INLINE_CHAIN = false>
#ifndef VG_EMIT_WAVES
#define VG_EMIT_WAVES 4    // waves per SIMD the register allocation of the emit kernels aims at (experiment switch, profiles/NOTES.md)
#endif
#ifndef VG_EMIT_MULTI_WAVES
#define VG_EMIT_MULTI_WAVES 4
#endif
__global__ __launch_bounds__(kEmitThreads) __attribute__((amdgpu_waves_per_eu(VG_EMIT_WAVES, 8))) void vg_emit_kernel(EmitArgs a)
{
    const unsigned int t = a.map_window ? xcd_window_block(blockIdx.x, gridDim.x, a.map_window) : xcd_contiguous_block(blockIdx.x, gridDim.x);
    emit_tile<MODEL, WANT_JAC, FRAMES_LDS, INLINE_CHAIN>(a, t * (unsigned)kEmitThreads);
}

// ------------------------------------------------------------------------------------------
// kernel 2, several datasets in ONE launch (stereo pair, camera rig): a problem's datasets are small launches each
// (2 000 stereo pairs: 42 + 62 MB) whose ramp-up, drain and inter-kernel gap cost a quarter of the pass.  The tiles of
// all datasets form one range; XCD x streams its share of EVERY dataset (the x-th eighth of its tiles, cut into runs of
// kEmitMapWindow tiles: xcd_window_block), dataset after dataset, so a die
// works inside one dataset's output arrays at a time and every die gets the same bytes and the same arithmetic whatever the
// mix of models (a rig's Mei tile writes 1.85 x the bytes of its UCM tile: with one contiguous piece of equal tile count
// per die, the dies holding the wide datasets finished last -- rig, 591 MB: 116 us against 109 us; profiles/NOTES.md "Merged
// emit launch").  The earlier cuts (one contiguous piece per die, of equal tile count or equal bytes: xcd_first / xcd_count)
// stay behind the emit_equal_tiles hook for A/B.  The grid is 8 x the longest piece; surplus workgroups leave at once.  Per-dataset arguments travel by value in the kernel argument segment (no table upload per
// evaluation); the camera model and the chain route are wave-uniform run-time switches over the same tile routine.
// ------------------------------------------------------------------------------------------
constexpr int kEmitMultiMax = 8;

struct EmitMultiArgs {
    EmitArgs ds[kEmitMultiMax];
    unsigned int first_tile[kEmitMultiMax + 1];
    int model[kEmitMultiMax];
    int inline_chain[kEmitMultiMax];
    int n;
    unsigned int xcd_first[8], xcd_count[8];   // tiles of XCD x: [xcd_first[x], xcd_first[x] + xcd_count[x])
    int per_dataset;                           // 1: XCD x takes the x-th eighth of EVERY dataset, dataset after dataset
};

template <int MODEL>
__device__ __forceinline__ void emit_tile_route(const EmitArgs &a, unsigned int o0, bool inline_chain)
{
    if (inline_chain) emit_tile<MODEL, true, true, true>(a, o0);
    else emit_tile<MODEL, true, true, false>(a, o0);
}

#ifdef VG_TU_CORE  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(kEmitThreads) __attribute__((amdgpu_waves_per_eu(VG_EMIT_MULTI_WAVES, 8))) void vg_emit_multi_kernel(EmitMultiArgs m)
{
    const unsigned int x = blockIdx.x & 7u;   // workgroup b runs on XCD b % 8 (observed dispatch order)
    unsigned int j = blockIdx.x >> 3, t;
    int d = 0;
    if (m.per_dataset) {
        for (;; d++) {
            if (d == m.n) return;
            const unsigned int nt = m.first_tile[d + 1] - m.first_tile[d], q = nt >> 3, r = nt & 7u, cnt = q + (x < r ? 1u : 0u);
            if (j < cnt) {
                const unsigned int W = m.ds[d].map_window;   // XCD x's j-th tile of this dataset
                t = m.first_tile[d] + (W ? xcd_window_block(j * 8u + x, nt, W) : x * q + (x < r ? x : r) + j);
                break;
            }
            j -= cnt;
        }
    } else {
        if (j >= m.xcd_count[x]) return;
        t = m.xcd_first[x] + j;
        while (d + 1 < m.n && t >= m.first_tile[d + 1]) d++;
    }
    const unsigned int o0 = (t - m.first_tile[d]) * (unsigned)kEmitThreads;
    const bool inl = m.inline_chain[d] != 0;
    switch (m.model[d]) {
    case kEUCM: emit_tile_route<kEUCM>(m.ds[d], o0, inl); break;
    case kUCM: emit_tile_route<kUCM>(m.ds[d], o0, inl); break;
    default: emit_tile_route<kMEI>(m.ds[d], o0, inl); break;
    }
}
#endif

// ------------------------------------------------------------------------------------------
// measurement helpers: pure streaming write / copy with the emit kernel's store pattern -- every wave
// instruction moves 1 KiB of consecutive bytes (16 B per lane) and a workgroup owns one contiguous
// 16 KiB run.  Used to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE and as the box's measured
// streaming rate.
// ------------------------------------------------------------------------------------------
constexpr int kStreamUnroll = 4;

#ifdef VG_TU_CORE  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(256) void vg_stream_write_kernel(double *__restrict__ dst, long long n2, double value)
{
    using d2 = HIP_vector_type<double, 2>;
    d2 v;
    v.x = value;
    v.y = value;
    d2 *d = reinterpret_cast<d2 *>(dst);
    const long long base = (long long)blockIdx.x * (256 * kStreamUnroll) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < kStreamUnroll; k++) {
        const long long i = base + k * 256;
        if (i < n2) d[i] = v;
    }
}
#endif

#ifdef VG_TU_CORE  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(256) void vg_stream_copy_kernel(double *__restrict__ dst, const double *__restrict__ src,
                                                              long long n2)
{
    using d2 = HIP_vector_type<double, 2>;
    d2 *d = reinterpret_cast<d2 *>(dst);
    const d2 *s = reinterpret_cast<const d2 *>(src);
    const long long base = (long long)blockIdx.x * (256 * kStreamUnroll) + threadIdx.x;
    d2 t[kStreamUnroll];
#pragma unroll
    for (int k = 0; k < kStreamUnroll; k++) {
        const long long i = base + k * 256;
        if (i < n2) t[k] = s[i];
    }
#pragma unroll
    for (int k = 0; k < kStreamUnroll; k++) {
        const long long i = base + k * 256;
        if (i < n2) d[i] = t[k];
    }
}
#endif

// What the FP64 vector pipe delivers on this box under the occupancy of the fused Gram kernels (two waves per SIMD: 256-thread
// workgroups, two per CU by their LDS reservation): every lane runs kFmaChains independent chains of dependent v_fma_f64 -- no
// memory, a loop of a few hundred bytes.  The guide's 78.6 TFLOP/s assume 2.4 GHz; under this load the part runs lower
// (profiles/NOTES.md, tools/exp/fp64_ramp.hip), and bench.py prints this next to the Gram kernel's roofline fraction.
constexpr int kFmaChains = 8;
#ifdef VG_TU_CORE  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(256) void vg_fp64_fma_kernel(double *__restrict__ out, int iters, double seed)
{
    extern __shared__ double fma_pad[];   // reserves the LDS that limits a CU to two workgroups; never touched
    double x[kFmaChains];
#pragma unroll
    for (int i = 0; i < kFmaChains; i++) x[i] = seed + 1e-3 * i + 1e-9 * threadIdx.x;
    const double a = 1.0000001, b = 1e-9;
#pragma unroll 4
    for (int k = 0; k < iters; k++)
#pragma unroll
        for (int i = 0; i < kFmaChains; i++) x[i] = __builtin_fma(x[i], a, b);
    double t = 0.;
#pragma unroll
    for (int i = 0; i < kFmaChains; i++) t += x[i];
    if (t == 12345.678) {   // never: keeps the chains alive
        fma_pad[threadIdx.x] = t;
        out[blockIdx.x] = fma_pad[threadIdx.x ^ 1];
    }
}
#endif

}  // namespace vg
