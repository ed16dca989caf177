// b4d_wiener_mr.hpp -- interface of the mixed-radix Wiener kernels (b4d_wiener_mr.hip) used by b4d_wiener.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace b4d {

struct WmrGeom {
    int h, w, py, px;   // frame and half kernel
    int H, W;           // padded size
    int Wh;             // W / 2 + 1 independent spectrum columns
    int Hp;             // pitch (complex words) of one column in the transposed workspace, a multiple of 16, > H if H is odd
    int hp;             // row pairs per frame, (H + 1) / 2
    int clip;
    float inv;          // 1 / (H W)
};

// the length has a compiled three-radix kernel
bool wmr_supported(int n);
// pitch (complex / float words) of one column of the transposed workspaces for column length H
inline int wmr_pitch(int H) { return (H + 1 + 15) / 16 * 16; }
// fft -> psd -> autocorr of real (nframes, ny, nx) frames in three passes + a zero-lag reduction (general sizes whose sides
// both have compiled kernels): T (nframes, nx/2 + 1, wmr_pitch(ny)) complex and Pt (same shape, float) are workspaces,
// scratch holds nframes * ((ny + 1) / 2 + 1) floats.  psd / autocorr: (nframes, ny, nx) float32, fftshift-ed, either may be
// null; flags: B4D_REMOVE_MEAN, B4D_NORM_PEAK.  signal/fft.py:261-309, signal/corr.py:256-320.
int wmr_psd_autocorr(const float* frames, int nframes, int ny, int nx, const float2* twx, const float2* twy, float2* T, float* Pt,
                     float* scratch, float* psd, float psd_scale, float* autocorr, unsigned flags, hipStream_t st);
// frames (nframes, h, w) -> T (nframes, Wh, Hp) transposed half spectra; pmax (nframes, hp) = max|rows of each pair|
int wmr_rows_fwd(const float* frames, float2* T, const float2* twx, float* pmax, const WmrGeom& g, int nframes, hipStream_t st);
// every column of T: forward transform, times filt (Wh, Hp), inverse transform (unscaled), in place;
// amax[f] = max|frame f| reduced from pmax
// sep_y != null: `filt` is not read; the filter of a separable, point-symmetric PSF is rebuilt per element from sep_x (Wh) =
// {hx[k], lx[k]} and sep_y (H) = {hy[ky], ly[ky]}: W = hx hy / ((hx hy)^2 + balance (lx + ly)^2)
int wmr_cols(float2* T, const float2* filt, const float2* twy, const float* pmax, float* amax, const WmrGeom& g, int nframes,
             hipStream_t st, const float2* sep_x = nullptr, const float2* sep_y = nullptr, float balance = 0.f);
// T -> out (nframes, h, w): inverse row transforms, 1/(H W), normalise by max|frame|, clip, rescale, crop
int wmr_rows_inv(const float2* T, float* out, const float2* twx, const float* amax, const WmrGeom& g, int nframes, hipStream_t st);

// ---- general-size correlation (tracking, xcorr2d) on the same passes.  Spectra / G workspaces hold wmr_spectrum_elems(ny, nx)
// complex words per item in the transposed [k][ky] layout.
size_t wmr_spectrum_elems(int ny, int nx);
int wmr_quads_per_frame(int ny);   // arg-max partials per map written by wmr_rows_magnitude
// real frames (nframes, ny, nx) -> 2-D half spectra S; scratch: nframes * (ny + 1) / 2 floats
int wmr_forward_spectra(const float* frames, int nframes, int ny, int nx, const float2* twx, const float2* twy, float2* S, float* scratch,
                        hipStream_t st);
// G[pair] = inverse column transforms of A[ia[pair]] conj(B[ib[pair]]) (whitened if `whiten`; DC zeroed with B4D_REMOVE_MEAN)
int wmr_product_inverse(const float2* A, const float2* B, const int* ia, const int* ib, int npairs, int ny, int nx, const float2* twy,
                        float2* G, int whiten, float eps, unsigned flags, hipStream_t st);
// G -> |corr| maps (npairs, ny, nx), fftshift-ed, scaled 1/(ny nx), + wmr_quads_per_frame(ny) arg-max partials per map.
// selw != null && pred_bin != 0: per map at selw + i * sel_stride, word 1 += values below the key bin pred_bin, word 2 += values
// in it, word 3 = append cursor of compact + i * ny * nx, which receives the bin's values (the median's first select step,
// b4d_track.hip; all zeroed by the caller)
int wmr_rows_magnitude(const float2* G, int npairs, int ny, int nx, const float2* twx, float* mag, float* part_val, int* part_idx,
                       unsigned* selw, int sel_stride, unsigned pred_bin, float* compact, hipStream_t st);
// fft2d of real frames: out (nframes, ny, nx) complex, fftshift-ed (signal/fft.py:198-237); S / scratch as wmr_forward_spectra
int wmr_fft2d(const float* frames, int nframes, int ny, int nx, const float2* twx, const float2* twy, float2* S, float* scratch,
              float2* out, hipStream_t st);
// G -> real maps (nframes, ny, nx), fftshift-ed, scaled 1/(ny nx)
int wmr_rows_real_out(const float2* G, int nframes, int ny, int nx, const float2* twx, float* out, hipStream_t st);

}  // namespace b4d
