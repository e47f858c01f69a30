// Host-side helpers of the vdmi C ABI: status strings, weight packing, relative-position buckets.
#include <cmath>
#include <cstring>
#include <vector>

#include "vdmi_common.h"

namespace {

inline uint16_t f32_to_bf16_rne(float f) { return vdmi_host_half_bits(f); }

inline float bf16_to_f32(uint16_t h) { return vdmi_host_half_value(h); }

}   // namespace

extern "C" const char* vdmi_strerror(int status) {
    switch (status) {
        case VDMI_OK: return "ok";
        case VDMI_E_ARG: return "invalid argument (null pointer, bad enum or inconsistent sizes)";
        case VDMI_E_SHAPE: return "shape not supported by the gfx950 kernels";
        case VDMI_E_ALIGN: return "pointer or leading dimension is not 16-byte aligned";
        case VDMI_E_LAUNCH: return "HIP kernel launch failed";
        case VDMI_E_LDS: return "tile does not fit the 160 KiB LDS";
        case VDMI_E_SCRATCH: return "input needs more than the library's static scratch buffer (use the entry point that takes a workspace)";
        default: return "unknown vdmi status";
    }
}

extern "C" int vdmi_version(void) { return 100; }

extern "C" size_t vdmi_pack_taps_elems(int n, int npad, int cin0, int cin1, int ntaps, int bkc) {
    (void)n;
    if (bkc <= 0) return 0;
    const size_t nch = (size_t)((cin0 + bkc - 1) / bkc) + (size_t)((cin1 + bkc - 1) / bkc);
    return nch * (size_t)ntaps * (size_t)npad * (size_t)bkc;
}

// Layout: [chunk][tap][npad/32][bkc/16][lane 0..63][j 0..7];
// element = W[cout = ct*32 + (lane & 31)][cin = chunk_base + ks*16 + 8*(lane >> 5) + j][tap]
// which is the A-operand fragment of v_mfma_f32_32x32x16_bf16 for rows = output channels.
extern "C" int vdmi_pack_taps(const float* w, int n, int npad, int cin0, int cin1, int ntaps, int bkc,
                              uint16_t* out, float* colsum) {
    if (!w || !out) return VDMI_E_ARG;
    if (n <= 0 || npad < n || npad % 32 || cin0 <= 0 || cin1 < 0 || ntaps <= 0 || (bkc != 32 && bkc != 64))
        return VDMI_E_SHAPE;
    const int cin = cin0 + cin1;
    const int nch0 = (cin0 + bkc - 1) / bkc, nch1 = (cin1 + bkc - 1) / bkc;
    const int ks_per = bkc / 16;
    std::vector<double> sums(colsum ? n : 0, 0.0);
    size_t o = 0;
    for (int c = 0; c < nch0 + nch1; ++c) {
        const bool first = c < nch0;
        const int cbase = first ? c * bkc : (c - nch0) * bkc;   // within the source
        const int clim = first ? cin0 : cin1;
        const int coff = first ? 0 : cin0;                      // into the concatenated cin axis
        for (int t = 0; t < ntaps; ++t)
            for (int ct = 0; ct < npad / 32; ++ct)
                for (int ks = 0; ks < ks_per; ++ks)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j, ++o) {
                            const int co = ct * 32 + (lane & 31);
                            const int ci = cbase + ks * 16 + 8 * (lane >> 5) + j;
                            uint16_t v = 0;
                            if (co < n && ci < clim) {
                                v = f32_to_bf16_rne(w[((size_t)co * cin + coff + ci) * ntaps + t]);
                                if (colsum) sums[co] += (double)bf16_to_f32(v);
                            }
                            out[o] = v;
                        }
    }
    if (colsum)
        for (int i = 0; i < n; ++i) colsum[i] = (float)sums[i];
    return VDMI_OK;
}

// RelativePositionBias._relative_position_bucket of the reference (vdp.py:286-304), evaluated for
// relative_position = j - i with the same float32 steps torch takes: log in float32, division by
// the float32-rounded log(max_distance / max_exact), multiplication, truncation to integer.
extern "C" int vdmi_relpos_buckets(int n, int num_buckets, int max_distance, int32_t* out) {
    if (!out || n <= 0 || num_buckets < 4) return VDMI_E_ARG;
    const int half = num_buckets / 2;
    const int max_exact = half / 2;
    const float denom = (float)std::log((double)max_distance / (double)max_exact);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            int nn = -(j - i);
            int ret = nn < 0 ? half : 0;
            nn = nn < 0 ? -nn : nn;
            int val;
            if (nn < max_exact) {
                val = nn;
            } else {
                const float r = std::log((float)nn / (float)max_exact) / denom * (float)(half - max_exact);
                val = max_exact + (int)r;
                if (val > half - 1) val = half - 1;
            }
            out[i * n + j] = ret + val;
        }
    return VDMI_OK;
}
