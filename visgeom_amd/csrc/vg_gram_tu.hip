// vg_gram_tu.hip -- translation unit of libvisgeom_amd.so: the normal-equation build (vg_dataset_gram_* / vg_problem_gram_*),
// i.e. the fused evaluate + Gram kernels of vg_gram_valu.hpp / vg_gram.hpp and their fixed-order sums.
// Built with hipcc for gfx950 only; compiled on its own so that an edit of a Gram kernel does not rebuild the emit kernels.
#define VG_TU_GRAM  // the non-template kernels this translation unit owns (the headers guard them by owner)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "vg_internal.hpp"
#include "vg_gram.hpp"
#include "vg_gram_valu.hpp"

using vgi::Camera;
using vgi::Dataset;
using vgi::fail;
using vgi::valid_dataset;

namespace {

void fill_gram_args(const vg_problem *p, const Dataset &d, vg::GramArgs &a, double *gram, const double *d_params)
{
    const Camera &cam = p->cams[d.camera];
    a.frames = d.d_frames;
    a.board = d.d_board;
    a.obs = d.d_obs;
    a.intr = d_params + cam.offset;
    a.res = nullptr;
#ifdef VG_GRAM_STAMPS
    a.res = reinterpret_cast<double *>(vgi::debug_hook(vgi::kHookGramStamps));   // measurement build: the clock stamps' buffer
#endif
    a.jac_intr = nullptr;
    for (int l = 0; l < vg::kMaxChain; l++) a.jac_member[l] = nullptr;
    a.gram = gram;
    a.n_blocks = (unsigned int)d.n_blocks;
    a.N = (unsigned int)d.N;
    a.L = d.L;
    a.W = cam.K + 6 * d.L + 1;
    a.frame_stride_d = d.frame_stride;
    a.gate = p->gram_gate;
    a.gate_expect = p->gram_gate_expect;
}

template <int MODEL>
int launch_gram_fused(hipStream_t stream, const vg::GramArgs &a)
{
    const bool rcol = a.W == 17;  // 16 Jacobian columns + residual (Mei mono): residual row / column on the lanes
    const int T = rcol ? 1 : (a.W + 15) / 16;
    // one LDS tile per wave (= per pair of images); as many waves per workgroup (<= 4) as fit in 80 KiB, so
    // that two workgroups share a CU (160 KiB LDS) and one can contract while the other evaluates
    const size_t tile = (size_t)vg::gram_wave_lds_doubles(a.W, a.frame_stride_d) * sizeof(double);
    int waves = (int)((80 * 1024) / tile);
    waves = waves < 1 ? 1 : (waves > vg::kGramMaxWavesPerBlock ? vg::kGramMaxWavesPerBlock : waves);
    const unsigned int n_pairs = (a.n_blocks + 1) / 2;
    const unsigned int grid = (n_pairs + waves - 1) / waves;
    const size_t lds = (size_t)waves * tile;
    const dim3 blk(waves * vg::kWave);
    if (rcol) hipLaunchKernelGGL((vg::vg_gram_fused_kernel<MODEL, 1, false, true>), dim3(grid), blk, lds, stream, a);
    else if (T == 1) hipLaunchKernelGGL((vg::vg_gram_fused_kernel<MODEL, 1>), dim3(grid), blk, lds, stream, a);
    else if (T == 2 && a.W - 16 <= vg::kCornerMax)
        hipLaunchKernelGGL((vg::vg_gram_fused_kernel<MODEL, 2, true>), dim3(grid), blk, lds, stream, a);
    else if (T == 2) hipLaunchKernelGGL((vg::vg_gram_fused_kernel<MODEL, 2>), dim3(grid), blk, lds, stream, a);
    else hipLaunchKernelGGL((vg::vg_gram_fused_kernel<MODEL, 3>), dim3(grid), blk, lds, stream, a);
    VG_HIP(hipGetLastError());
    return VG_OK;
}


// Narrow row blocks (W <= 13, chain of at most one member) take the vector-pipe kernel of vg_gram_valu.hpp; a single
// DIRECT member is walked in-kernel.  Both are pure functions of the problem (never of call history).
bool gram_uses_valu(const vg_problem *p, const Dataset &d)
{
    const bool force_mfma = vgi::debug_hook(vgi::kHookGramForceMfma) != 0;  // measurement hook (A/B of the two kernels)
    (void)p;
    return !force_mfma && d.L <= vg::kMaxChain;  // one member: the direct form; two or more: the factored form (vg_gram_valu_z_kernel)
}

bool gram_inline_chain(const vg_problem *p, const Dataset &d)
{
    return !p->force_prepared_frames && gram_uses_valu(p, d) && d.L == 1 && d.status[0] == VG_TRANSFORM_DIRECT;
}

// The persistent form (vg_gram_valu_pers_kernel): a single DIRECT member walked in the kernel, blocks up to 13 wide, a board of
// exactly 96 points (8 x 12), at least one full round of the one-shot kernel's workgroups.  Returns the workgroups it runs
// with (= the number of partials it leaves; 0 = does not apply) and the shape (threads per workgroup).
template <int MODEL, int CH, int THREADS>
unsigned int gram_pers_resident()
{
    constexpr int W = vg::CameraTraits<MODEL>::K + 7;
    static unsigned int resident = 0;   // per instantiation; the device's properties do not change
    if (!resident) {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        constexpr size_t lds = vg::gram_valu_pers_lds_bytes<W, THREADS>();
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(vg::vg_gram_valu_pers_kernel<MODEL, CH, THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, vg::vg_gram_valu_pers_kernel<MODEL, CH, THREADS>, THREADS, lds) != hipSuccess ||
            hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || per_cu < 1)
            return 0;
        resident = (unsigned int)per_cu * (unsigned int)prop.multiProcessorCount;
    }
    return resident;
}

// which shape, by size: none below 4 096 images (less than a round of the one-shot kernel's octets), the eight-wave shape up to
// 16 384, the four-wave shape beyond (profiles/r05z_gram_pers_probe5.txt; NOTES).  hook gram_persistent: 1 = never,
// 2 = the four-wave shape whenever it applies, 3 = the eight-wave shape whenever it applies
// the persistent kernel leaves one partial sum per workgroup: its grid never exceeds this, and the partial buffer is sized from the
// same constant (the occupancy query is at most ~608 on this part, but nothing tied the two together)
constexpr unsigned int kPersMaxWorkgroups = 1024;

// doubles of a dataset's per-workgroup partial sums, whichever Gram kernel writes them: one per octet of images (one-shot kernel)
// or one per workgroup of the persistent grid (at most one per image pair, never more than kPersMaxWorkgroups)
inline size_t gram_partial_count(int64_t n_blocks)
{
    const size_t octets = (size_t)((n_blocks + vg::kValuImagesPerBlock - 1) / vg::kValuImagesPerBlock), n_pairs = ((size_t)n_blocks + 1) / 2;
    return std::max<size_t>(octets, std::min<size_t>(n_pairs, kPersMaxWorkgroups));
}

inline int gram_pers_shape(const vg::GramValuArgs &a, bool with_sum)
{
    const long long hook = vgi::debug_hook(vgi::kHookGramPersistent);
    if (hook == 1 || a.g.N != (unsigned)(vg::kValuLanesPerImage * vg::kPersCorners) || a.g.n_blocks < 2) return 0;   // three corners per lane: the 8 x 12 board
    if (hook == 2) return vg::kPersThreadsLong;
    if (hook == 3) return vg::kPersThreadsShort;
    if (a.g.n_blocks >= 16384u) return vg::kPersThreadsLong;
    // below: the eight-wave shape ties with the one-shot kernel as a launch (10 k images: 19.2 / 19.0 us) and wins through the 256
    // partials it leaves for the sum (22.7 -> 21.75 us): only when the sum is asked for
    return (with_sum && a.g.n_blocks >= 4096u) ? vg::kPersThreadsShort : 0;
}

template <int MODEL, int CH, int THREADS>
int launch_gram_valu_pers(hipStream_t stream, const vg::GramValuArgs &a, unsigned int *n_wg_out)
{
    constexpr int W = vg::CameraTraits<MODEL>::K + 7;
    const unsigned int resident = gram_pers_resident<MODEL, CH, THREADS>();
    if (!resident) return fail(VG_ERR_HIP, "occupancy query of the persistent Gram kernel failed");
    const unsigned int n_pairs = (a.g.n_blocks + 1) / 2, n_wg = std::min(std::min(n_pairs, resident), kPersMaxWorkgroups);
    vg::GramValuArgs ap = a;
    ap.n_wg = n_wg;
    constexpr size_t lds = vg::gram_valu_pers_lds_bytes<W, THREADS>();
    hipLaunchKernelGGL((vg::vg_gram_valu_pers_kernel<MODEL, CH, THREADS>), dim3(n_wg), dim3(THREADS), lds, stream, ap, n_pairs);
    VG_HIP(hipGetLastError());
    *n_wg_out = n_wg;
    return VG_OK;
}

template <int MODEL, int CH>
int launch_gram_valu_pers_shape(hipStream_t stream, const vg::GramValuArgs &a, int shape, unsigned int *n_wg_out)
{
    return shape == vg::kPersThreadsShort ? launch_gram_valu_pers<MODEL, CH, vg::kPersThreadsShort>(stream, a, n_wg_out)
                                          : launch_gram_valu_pers<MODEL, CH, vg::kPersThreadsLong>(stream, a, n_wg_out);
}

template <int MODEL, int L, int CH>
int launch_gram_valu_lch(hipStream_t stream, const vg::GramValuArgs &a, bool inline_chain)
{
    const dim3 grid(a.n_wg), blk(vg::kValuThreads);
    const size_t lds = vg::gram_valu_lds_bytes(vg::CameraTraits<MODEL>::K + 6 * L + 1, L);
    if constexpr (L == 1) {
        if (inline_chain) hipLaunchKernelGGL((vg::vg_gram_valu_kernel<MODEL, 1, true, CH>), grid, blk, lds, stream, a);
        else hipLaunchKernelGGL((vg::vg_gram_valu_kernel<MODEL, 1, false, CH>), grid, blk, lds, stream, a);
    } else {
        hipLaunchKernelGGL((vg::vg_gram_valu_kernel<MODEL, L, false, CH>), grid, blk, lds, stream, a);
    }
    VG_HIP(hipGetLastError());
    return VG_OK;
}

// corners per lane in a full chunk, by the width of the row block (register file): three up to 13 columns (an 8 x 12 board
// is one chunk), two up to 19, one beyond; boards of at most one corner per lane of the half-wave never need more than one
template <int MODEL, int L>
int launch_gram_valu_l(hipStream_t stream, const vg::GramValuArgs &a, bool inline_chain)
{
    const bool force_ch1 = vgi::debug_hook(vgi::kHookGramCh1) != 0;  // measurement hook
    constexpr int W = vg::CameraTraits<MODEL>::K + 6 * L + 1;
#ifdef VG_GRAM_CH2
    constexpr int kMain = W <= 19 ? 2 : 1;   // tools/exp A/B build: two corners per lane, three waves per SIMD on the 13-wide blocks
#else
    constexpr int kMain = W <= 13 ? 3 : (W <= 19 ? 2 : 1);
#endif
    if constexpr (kMain > 1)
        if (!force_ch1 && a.g.N > (unsigned)vg::kValuLanesPerImage) return launch_gram_valu_lch<MODEL, L, kMain>(stream, a, inline_chain);
    return launch_gram_valu_lch<MODEL, L, 1>(stream, a, inline_chain);
}

template <int MODEL>
int launch_gram_valu(hipStream_t stream, const vg::GramValuArgs &a, int L, bool inline_chain)
{
    if (L == 0) return launch_gram_valu_l<MODEL, 0>(stream, a, false);
    if (L == 1) return launch_gram_valu_l<MODEL, 1>(stream, a, inline_chain);
    // two or more members: the factored form -- rows of K + 7 columns whatever L is
    constexpr int K = vg::CameraTraits<MODEL>::K;
    constexpr int CH = K + 7 <= 13 ? 3 : 2;
    const size_t lds = vg::gram_valu_z_lds_bytes(K, L);
    const bool small_board = a.g.N <= (unsigned)vg::kValuLanesPerImage || vgi::debug_hook(vgi::kHookGramCh1) != 0;
    if (lds > 48 * 1024) {  // chains of four or five members: more than the default dynamic LDS limit
        const void *fn = small_board ? reinterpret_cast<const void *>(vg::vg_gram_valu_z_kernel<MODEL, 1>)
                                     : reinterpret_cast<const void *>(vg::vg_gram_valu_z_kernel<MODEL, CH>);
        VG_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (small_board) hipLaunchKernelGGL((vg::vg_gram_valu_z_kernel<MODEL, 1>), dim3(a.n_wg), dim3(vg::kValuThreads), lds, stream, a);
    else hipLaunchKernelGGL((vg::vg_gram_valu_z_kernel<MODEL, CH>), dim3(a.n_wg), dim3(vg::kValuThreads), lds, stream, a);
    VG_HIP(hipGetLastError());
    return VG_OK;
}

}  // namespace

bool vgi::gram_dataset_needs_frames(const vg_problem *p, int dataset_id)
{
    const Dataset &d = p->dss[(size_t)dataset_id];
    return d.n_blocks && !gram_inline_chain(p, d);
}

bool vgi::gram_needs_frames(const vg_problem *p)
{
    for (const Dataset &d : p->dss)
        if (d.n_blocks && !gram_inline_chain(p, d)) return true;
    return false;
}

int vgi::gram_fused_at(vg_problem *p, int dataset_id, const double *d_params, double *gram, double *sum)
{
    Dataset &d = p->dss[dataset_id];
    const Camera &cam = p->cams[d.camera];
    const int W = cam.K + 6 * d.L + 1;
    if (!d.n_blocks) {
        if (sum) VG_HIP(hipMemsetAsync(sum, 0, sizeof(double) * W * W, p->stream));
        return VG_OK;
    }
    if (!gram) return fail(VG_ERR_INVALID_ARGUMENT, "gram is NULL");
    if (d.n_blocks > 0x7fffffff) return fail(VG_ERR_INVALID_ARGUMENT, "too many blocks for one launch");
    int rc;
    if (gram_uses_valu(p, d)) {
        vg::GramValuArgs a;
        fill_gram_args(p, d, a.g, gram, d_params);
        const bool inl = gram_inline_chain(p, d);
        a.chain_params = d.L ? d_params + d.chain.base[0] : nullptr;
        a.chain_stride = d.L ? d.chain.stride[0] : 0;
        a.seq_index = d.seq_identity ? nullptr : d.d_seq;
        a.n_wg = (unsigned int)((d.n_blocks + vg::kValuImagesPerBlock - 1) / vg::kValuImagesPerBlock);
        a.partials = nullptr;
        const int E = W * (W + 1) / 2;
        if (sum) {
            // room for either kernel's partials: one per octet (one-shot), one per resident workgroup (persistent: at most one per
            // image pair and never more than 1 024)
            const size_t n_part = gram_partial_count(d.n_blocks);
            if (!d.d_wg_partials) VG_HIP(hipMalloc(&d.d_wg_partials, sizeof(double) * (size_t)E * n_part));
            a.partials = d.d_wg_partials;
        }
        const int pers = (inl && d.L == 1) ? gram_pers_shape(a, sum != nullptr) : 0;
        if (pers) {
            unsigned int n_wg = 0;
            switch (cam.model) {
            case VG_MODEL_EUCM: rc = launch_gram_valu_pers_shape<vg::kEUCM, 3>(p->stream, a, pers, &n_wg); break;
            case VG_MODEL_UCM: rc = launch_gram_valu_pers_shape<vg::kUCM, 3>(p->stream, a, pers, &n_wg); break;
            default: rc = launch_gram_valu_pers_shape<vg::kMEI, 2>(p->stream, a, pers, &n_wg); break;   // 17-wide rows: two corners, then the third
            }
            a.n_wg = n_wg;
        } else {
            switch (cam.model) {
            case VG_MODEL_EUCM: rc = launch_gram_valu<vg::kEUCM>(p->stream, a, d.L, inl); break;
            case VG_MODEL_UCM: rc = launch_gram_valu<vg::kUCM>(p->stream, a, d.L, inl); break;
            default: rc = launch_gram_valu<vg::kMEI>(p->stream, a, d.L, inl); break;
            }
        }
        if (rc != VG_OK || !sum) return rc;
        hipLaunchKernelGGL(vg::vg_gram_partials_sum_kernel, dim3(E), dim3(256), 0, p->stream,
                           (const double *)d.d_wg_partials, a.n_wg, W, sum);
        VG_HIP(hipGetLastError());
        return VG_OK;
    }
    vg::GramArgs a;
    fill_gram_args(p, d, a, gram, d_params);
    switch (cam.model) {
    case VG_MODEL_EUCM: rc = launch_gram_fused<vg::kEUCM>(p->stream, a); break;
    case VG_MODEL_UCM: rc = launch_gram_fused<vg::kUCM>(p->stream, a); break;
    default: rc = launch_gram_fused<vg::kMEI>(p->stream, a); break;
    }
    if (rc != VG_OK || !sum) return rc;
    return vgi::gram_sum_into(p, dataset_id, gram, sum);
}

bool vgi::gram_merge_covers_all(const vg_problem *p)
{
    if (vgi::debug_hook(vgi::kHookGramNoMerge) || vgi::debug_hook(vgi::kHookGramCh1)) return false;
    int n = 0;
    for (const Dataset &d : p->dss) {
        if (!d.n_blocks) continue;
        if (!(d.n_blocks <= 0x7fffffff && gram_uses_valu(p, d) && d.L >= 1 && d.N > vg::kValuLanesPerImage)) return false;
        n++;
    }
    return n >= 2 && n % vg::kGramMultiMax != 1;  // a lone leftover group would go the ordinary way
}

int vgi::gram_fused_merged_at(vg_problem *p, const double *d_params, double *const *grams, std::vector<char> &taken,
                              double *const *partials)
{
    const bool off = vgi::debug_hook(vgi::kHookGramNoMerge) == 1 || vgi::debug_hook(vgi::kHookGramCh1) != 0;  // measurement hooks
    const int n_ds = (int)p->dss.size();
    taken.assign((size_t)n_ds, 0);
    std::vector<int> ids;
    for (int i = 0; i < n_ds && !off; i++) {
        const Dataset &d = p->dss[i];
        if (d.n_blocks > 0 && d.n_blocks <= 0x7fffffff && grams[i] && gram_uses_valu(p, d) && d.L >= 1 && d.N > vg::kValuLanesPerImage)
            ids.push_back(i);
    }
    if (ids.size() < 2) return VG_OK;
    // heaviest workgroups first: the launch ends on the light datasets' workgroups instead of a tail of the widest blocks
    // (same box, merged launch in dataset order -> heaviest first: stereo 14.4 -> 12.7 us, rig 66-71 -> 61-64 us)
    if (vgi::debug_hook(vgi::kHookGramNoMerge) != 2)
        std::stable_sort(ids.begin(), ids.end(), [&](int a2, int b2) {
            const Dataset &da = p->dss[a2], &db = p->dss[b2];
            return p->cams[da.camera].K + 6 * da.L > p->cams[db.camera].K + 6 * db.L;
        });
    for (size_t g0 = 0; g0 < ids.size(); g0 += vg::kGramMultiMax) {
        vg::GramValuMultiArgs m;
        m.n = (int)(ids.size() - g0 < (size_t)vg::kGramMultiMax ? ids.size() - g0 : (size_t)vg::kGramMultiMax);
        if (m.n < 2) break;  // a lone leftover goes the ordinary way
        unsigned int wgs = 0;
        size_t lds = 0;
        for (int k = 0; k < m.n; k++) {
            const Dataset &d = p->dss[ids[g0 + k]];
            const Camera &cam = p->cams[d.camera];
            vg::GramValuArgs &a = m.ds[k];
            fill_gram_args(p, d, a.g, grams[ids[g0 + k]], d_params);
            a.chain_params = d_params + d.chain.base[0];
            a.chain_stride = d.chain.stride[0];
            a.seq_index = d.seq_identity ? nullptr : d.d_seq;
            a.n_wg = (unsigned int)((d.n_blocks + vg::kValuImagesPerBlock - 1) / vg::kValuImagesPerBlock);
            a.partials = partials ? partials[ids[g0 + k]] : nullptr;
            m.kind[k] = 3 * cam.model + (d.L >= 2 ? 2 : (gram_inline_chain(p, d) ? 0 : 1));
            m.first_wg[k] = wgs;
            wgs += a.n_wg;
            const size_t need = d.L >= 2 ? vg::gram_valu_z_lds_bytes(cam.K, d.L) : vg::gram_valu_lds_bytes(cam.K + 6 * d.L + 1, d.L);
            lds = need > lds ? need : lds;
            taken[(size_t)ids[g0 + k]] = 1;
        }
        for (int k = m.n; k <= vg::kGramMultiMax; k++) m.first_wg[k] = wgs;
        if (lds > 48 * 1024)
            VG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(vg::vg_gram_valu_multi_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(vg::vg_gram_valu_multi_kernel, dim3(wgs), dim3(vg::kValuThreads), lds, p->stream, m);
        VG_HIP(hipGetLastError());
    }
    return VG_OK;
}

extern "C" {

int vg_dataset_gram_fused(vg_problem *p, int dataset_id, double *gram)
{
    int rc = valid_dataset(p, dataset_id);
    if (rc != VG_OK) return rc;
    if (!p->finalized) return fail(VG_ERR_STATE, "problem not finalized");
    VG_HIP(hipSetDevice(p->device));
    if (!gram_inline_chain(p, p->dss[dataset_id]) && (rc = vgi::ensure_frames(p)) != VG_OK) return rc;
    return vgi::gram_fused_at(p, dataset_id, p->d_params, gram, nullptr);
}

int vg_problem_gram_fused(vg_problem *p, double *const *grams)
{
    if (!p || !grams) return fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!p->finalized) return fail(VG_ERR_STATE, "problem not finalized");
    VG_HIP(hipSetDevice(p->device));
    int rc;
    for (size_t i = 0; i < p->dss.size(); i++)
        if (p->dss[i].n_blocks && !grams[i]) return fail(VG_ERR_INVALID_ARGUMENT, "gram is NULL");
    if (vgi::gram_needs_frames(p) && (rc = vgi::ensure_frames(p)) != VG_OK) return rc;
    std::vector<char> taken;
    if ((rc = vgi::gram_fused_merged_at(p, p->d_params, grams, taken, nullptr)) != VG_OK) return rc;
    for (size_t i = 0; i < p->dss.size(); i++)
        if (!taken[i] && p->dss[i].n_blocks && (rc = vgi::gram_fused_at(p, (int)i, p->d_params, grams[i], nullptr)) != VG_OK) return rc;
    return VG_OK;
}

int vg_problem_gram_fused_sum(vg_problem *p, double *const *grams, double *const *sums)
{
    if (!p || !grams || !sums) return fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!p->finalized) return fail(VG_ERR_STATE, "problem not finalized");
    VG_HIP(hipSetDevice(p->device));
    const int n_ds = (int)p->dss.size();
    int rc;
    for (int i = 0; i < n_ds; i++) {
        if (!sums[i]) return fail(VG_ERR_INVALID_ARGUMENT, "sum is NULL");
        if (p->dss[i].n_blocks && !grams[i]) return fail(VG_ERR_INVALID_ARGUMENT, "gram is NULL");
    }
    if (vgi::gram_needs_frames(p) && (rc = vgi::ensure_frames(p)) != VG_OK) return rc;
    // the datasets that share the merged launch leave per-workgroup partial sums; ONE launch adds them for all of them
    std::vector<double *> parts((size_t)n_ds, nullptr);
    for (int i = 0; i < n_ds; i++) {
        Dataset &d = p->dss[i];
        if (!d.n_blocks || d.n_blocks > 0x7fffffff) continue;
        const int W = p->cams[d.camera].K + 6 * d.L + 1, E = W * (W + 1) / 2;
        const size_t n_part = gram_partial_count(d.n_blocks);   // as in gram_fused_at: the buffer is shared
        if (!d.d_wg_partials) VG_HIP(hipMalloc(&d.d_wg_partials, sizeof(double) * (size_t)E * n_part));
        parts[(size_t)i] = d.d_wg_partials;
    }
    std::vector<char> taken;
    if ((rc = vgi::gram_fused_merged_at(p, p->d_params, grams, taken, parts.data())) != VG_OK) return rc;
    vg::PartialSumArgs a;
    a.n = 0;
    unsigned int blocks = 0;
    auto flush = [&]() {
        if (!a.n) return;
        hipLaunchKernelGGL(vg::vg_gram_partials_sum_args_kernel, dim3(blocks), dim3(256), 0, p->stream, a);
        a.n = 0;
        blocks = 0;
    };
    for (int i = 0; i < n_ds; i++) {
        if (!taken[(size_t)i]) continue;
        const Dataset &d = p->dss[i];
        const int W = p->cams[d.camera].K + 6 * d.L + 1;
        vg::PartialSumDataset &pd = a.ds[a.n++];
        pd.partials = parts[(size_t)i];
        pd.out = sums[i];
        pd.n_wg = (unsigned int)((d.n_blocks + vg::kValuImagesPerBlock - 1) / vg::kValuImagesPerBlock);
        pd.W = W;
        pd.first_block = blocks;
        blocks += (unsigned int)(W * (W + 1) / 2);
        if (a.n == vg::kPartialSumMax) flush();
    }
    flush();
    VG_HIP(hipGetLastError());
    for (int i = 0; i < n_ds; i++)   // what the merged launch did not take (a single dataset, tiny boards, empty datasets)
        if (!taken[(size_t)i] && (rc = vgi::gram_fused_at(p, i, p->d_params, grams[i], sums[i])) != VG_OK) return rc;
    return VG_OK;
}

int vg_dataset_gram_fused_sum(vg_problem *p, int dataset_id, double *gram, double *sum)
{
    int rc = valid_dataset(p, dataset_id);
    if (rc != VG_OK) return rc;
    if (!p->finalized) return fail(VG_ERR_STATE, "problem not finalized");
    if (!sum) return fail(VG_ERR_INVALID_ARGUMENT, "sum is NULL");
    VG_HIP(hipSetDevice(p->device));
    if (!gram_inline_chain(p, p->dss[dataset_id]) && (rc = vgi::ensure_frames(p)) != VG_OK) return rc;
    return vgi::gram_fused_at(p, dataset_id, p->d_params, gram, sum);
}

int vg_dataset_gram_from_rows(vg_problem *p, int dataset_id, const double *residuals, const double *jac_intr,
                              const double *const *jac_member, double *gram)
{
    int rc = valid_dataset(p, dataset_id);
    if (rc != VG_OK) return rc;
    if (!p->finalized) return fail(VG_ERR_STATE, "problem not finalized");
    Dataset &d = p->dss[dataset_id];
    if (!d.n_blocks) return VG_OK;
    if (!gram || !residuals || !jac_intr || (d.L > 0 && !jac_member))
        return fail(VG_ERR_INVALID_ARGUMENT, "two-pass Gram needs residuals and every Jacobian block");
    for (int l = 0; l < d.L; l++)
        if (!jac_member[l]) return fail(VG_ERR_INVALID_ARGUMENT, "two-pass Gram needs residuals and every Jacobian block");
    if (d.n_blocks > 0x7fffffff) return fail(VG_ERR_INVALID_ARGUMENT, "too many blocks for one launch");
    VG_HIP(hipSetDevice(p->device));
    vg::GramArgs a;
    fill_gram_args(p, d, a, gram, p->d_params);
    a.res = residuals;
    a.jac_intr = jac_intr;
    for (int l = 0; l < d.L; l++) a.jac_member[l] = jac_member[l];
    const int K = p->cams[d.camera].K;
    const unsigned int n_pairs = (a.n_blocks + 1) / 2;
    const unsigned int grid = (n_pairs + vg::kGramMaxWavesPerBlock - 1) / vg::kGramMaxWavesPerBlock;
    const dim3 blk(vg::kGramMaxWavesPerBlock * vg::kWave);
    const int T = (a.W + 15) / 16;
    if (T == 1) hipLaunchKernelGGL((vg::vg_gram_rows_kernel<1>), dim3(grid), blk, 0, p->stream, a, K);
    else if (T == 2) hipLaunchKernelGGL((vg::vg_gram_rows_kernel<2>), dim3(grid), blk, 0, p->stream, a, K);
    else hipLaunchKernelGGL((vg::vg_gram_rows_kernel<3>), dim3(grid), blk, 0, p->stream, a, K);
    VG_HIP(hipGetLastError());
    return VG_OK;
}

}  // extern "C"

int vgi::gram_sum_into(vg_problem *p, int dataset_id, const double *gram, double *sum)
{
    Dataset &d = p->dss[dataset_id];
    const int W = p->cams[d.camera].K + 6 * d.L + 1;
    const int entries = W * W;
    if (!d.n_blocks) {
        VG_HIP(hipMemsetAsync(sum, 0, sizeof(double) * entries, p->stream));
        return VG_OK;
    }
    if (!gram) return fail(VG_ERR_INVALID_ARGUMENT, "gram is NULL");
    const unsigned int n = (unsigned int)d.n_blocks;
    const unsigned int parts = (n + vg::kSlab - 1) / vg::kSlab;
    if (!d.d_partials) VG_HIP(hipMalloc(&d.d_partials, sizeof(double) * (size_t)parts * entries));
    hipLaunchKernelGGL(vg::vg_gram_slab_sum_kernel, dim3(parts), dim3(256), 0, p->stream, gram, n, entries, d.d_partials);
    VG_HIP(hipGetLastError());
    hipLaunchKernelGGL(vg::vg_gram_final_sum_kernel, dim3((entries + 3) / 4), dim3(256), 0, p->stream,
                       (const double *)d.d_partials, parts, entries, sum);
    VG_HIP(hipGetLastError());
    return VG_OK;
}

extern "C" {

int vg_dataset_gram_sum(vg_problem *p, int dataset_id, const double *gram, double *sum)
{
    int rc = valid_dataset(p, dataset_id);
    if (rc != VG_OK) return rc;
    if (!p->finalized) return fail(VG_ERR_STATE, "problem not finalized");
    if (!sum) return fail(VG_ERR_INVALID_ARGUMENT, "sum is NULL");
    VG_HIP(hipSetDevice(p->device));
    return vgi::gram_sum_into(p, dataset_id, gram, sum);
}

}  // extern "C"
