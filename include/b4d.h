/* b4d.h -- C ABI of the MI355X (gfx950) hot path of barc4dip.
 *
 * The reference (barc4dip, pure Python) has no FFI: its boundary is the Python API of
 * barc4dip.signal / barc4dip.metrics / barc4dip.preprocessing.  This header is what a
 * maintainer binds (ctypes) to route those functions to the GPU.
 * Each entry point cites the reference function it serves (paths under
 * /root/reference/src/barc4dip).
 *
 * Conventions
 *   - every function returns 0 on success, a negative B4D_E* code on failure; the message
 *     is available from b4d_last_error() (thread local).  No C++ exceptions cross the ABI.
 *   - all data pointers are DEVICE pointers owned by the caller (e.g. tensor.data_ptr());
 *     images are row-major (ny, nx) float32, stacks (batch, ny, nx).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are
 *     asynchronous on that stream; a plan may be used from one stream at a time.
 *   - FFT outputs are fftshift-ed (DC at [ny/2, nx/2]) exactly as signal/fft.py:7-10 and
 *     signal/corr.py:7-10 define.
 */
#ifndef B4D_H
#define B4D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define B4D_OK 0
#define B4D_EINVAL (-1)   /* bad argument (null pointer, batch <= 0, ...) */
#define B4D_ESIZE (-2)    /* (ny, nx) not supported by the compiled kernels */
#define B4D_EHIP (-3)     /* a HIP runtime call failed */
#define B4D_ENOMEM (-4)

/* flags for b4d_autocorr2 / b4d_psd_autocorr2 (signal/corr.py:169-180 keyword arguments) */
#define B4D_REMOVE_MEAN 1u      /* remove_mean=True  : zero the DC bin of the power spectrum */
#define B4D_NORM_PEAK 2u        /* normalize="peak"  : divide by the zero-lag value, peak == 1 */
#define B4D_STANDARDIZE 4u      /* standardize=True  : divide by the variance (only visible with normalize="none") */

typedef struct b4d_plan b4d_plan;

/* Library / device ------------------------------------------------------------------- */
const char* b4d_version(void);
const char* b4d_last_error(void);
/* Process-wide switches that never change results, only the route taken (tests run every route):
 *   "track_predict_bin"  0 / 1 / 2 (default 1): 1 = b4d_phase_correlation counts and gathers the EXPECTED median bin of every
 *                        correlation map in the pass that produces it and (power-of-two sizes) does not store the map: the rows
 *                        around the peak are recomputed for the sub-pixel step, pairs whose expectation fails get their full map
 *                        from a second, gated pass; 0 = no expectation: full maps, full select; 2 = a deliberately wrong
 *                        expectation (test hook: every pair takes the gated pass).
 *   "lanes"              0 / 1 (default 1): 1 = the multi-pass entry points (b4d_fft2d, the general-size passes, b4d_wiener_apply,
 *                        b4d_phase_correlation) cut a stack into cache-sized launch groups and deal them alternately to the
 *                        caller's stream and ONE library-owned stream per device, forked from and joined into the caller's stream
 *                        by events inside the call (stream order as seen by the caller is unchanged); 0 = everything on the
 *                        caller's stream alone (for callers that must not see a second stream).
 *   "exp"                0 .. 255 (default 0): development switch for A/B runs of kernel variants under test in ONE process
 *                        (tools/dev_*.py); a shipped library has no reader of it.
 * Values outside an option's range and unknown names return B4D_EINVAL; the options are atomics, read once per entry-point call. */
int b4d_set_option(const char* name, int value);
/* 1 if (ny, nx) has a plan: powers of two in [64, 4096] (radix FFT kernels); any sides <= 512 (DFT-matrix products);
 * sides <= 8192 that split as 2^k * A * B with A + B <= 128 (fused in-LDS mixed radix) or any other side <= 4096
 * (Bluestein over that transform), at most 2^26 pixels. */
int b4d_size_supported(int ny, int nx);

/* Plans ------------------------------------------------------------------------------
 * A plan owns the twiddle tables and a workspace of `chunk` half-spectra
 * (chunk * ny * nx/2 complex64; general-length plans: three full complex buffers).  Batches larger
 * than `chunk` are processed chunk by chunk.                                            */
int b4d_plan_create(int ny, int nx, int chunk, b4d_plan** out);
int b4d_plan_destroy(b4d_plan* plan);
size_t b4d_plan_workspace_bytes(const b4d_plan* plan);

/* signal/fft.py:198-237 fft2d -- F = fftshift(fft2(img)) for a batch of real frames.
 * out: (batch, ny, nx) complex64 interleaved (re, im).                                  */
int b4d_fft2d(b4d_plan* plan, const float* frames, int batch, float* out_c64, void* stream);

/* Complex frames (signal/fft.py:198-258 accept complex input): plans from b4d_plan_create_general run the
 * general-length engines (DFT matrices up to 512, fused mixed radix beyond) for every size, powers of two included.
 * inverse == 0: out = fftshift(fft2(in));  inverse != 0: out = ifft2(ifftshift(in)), i.e. ifft2d of a shifted spectrum.
 * in / out: DEVICE (batch, ny, nx) complex64 (interleaved float pairs).                                               */
int b4d_plan_create_general(int ny, int nx, int chunk, b4d_plan** out);
int b4d_fft2d_c2c(b4d_plan* plan, const float* in_c64, int batch, int inverse, float* out_c64, void* stream);

/* signal/fft.py:261-309 psd2d -- P = |fftshift(fft2(img))|^2 * scale, scale = dx*dy/(nx*ny)
 * when scale=True else 1.  psd: (batch, ny, nx) float32.                                */
int b4d_psd2d(b4d_plan* plan, const float* frames, int batch, float* psd, float scale, void* stream);

/* signal/corr.py:256-320 autocorr2d -- circular autocorrelation, shifted, float32.
 * With B4D_NORM_PEAK the zero-lag sample is exactly 1.0f.                               */
int b4d_autocorr2d(b4d_plan* plan, const float* frames, int batch, float* autocorr, unsigned flags,
                   void* stream);

/* The north-star pipeline: one forward transform serves psd2d and
 * autocorr2d.  Either output pointer may be NULL.                                       */
int b4d_psd_autocorr2d(b4d_plan* plan, const float* frames, int batch, float* psd, float psd_scale,
                       float* autocorr, unsigned flags, void* stream);

/* Same call, but every kernel launch is bracketed by HIP events on `stream`; the elapsed times
 * (ms) of {row R2C, column FFT/PSD/inverse, zero-lag peak, row C2R} are ADDED to kernel_ms[0..3].
 * Synchronises the stream before returning.  Measurement aid for bench.py's roofline block.    */
int b4d_psd_autocorr2d_timed(b4d_plan* plan, const float* frames, int batch, float* psd, float psd_scale,
                             float* autocorr, unsigned flags, void* stream, float* kernel_ms);

/* Workspace placement (power-of-two plans; a no-op returning B4D_OK on the others).  Where a multi-GB allocation lands in
 * device memory decides 5-10 % of the time of every kernel that streams through it -- a property of the allocation that
 * stays with it until it is freed (same code, same virtual layout, two hipMalloc's of one process).
 * The plan owns its half-spectrum workspace, so it can measure: up to `candidates` - 1 (at most 7) further workspaces are
 * allocated one after the other, the caller's own b4d_psd_autocorr2d call is timed with each (two passes after an untimed
 * one), the fastest stays with the plan and the others are freed.  An allocation failure only ends the search.  Synchronises
 * `stream`; on return psd / autocorr hold the result of a normal call.  best_ms / worst_ms (optional) receive the time per
 * pass on the kept and on the slowest candidate.  All candidates are alive until the choice is made: up to
 * (candidates - 1) x b4d_plan_workspace_bytes of extra device memory for the duration of the call.
 * A diagnostic since round 3: five consecutive processes on one box ran the untuned plan within 0.3 % of the tuned one
 * (profiles/r03_placement.txt), and bench.py no longer calls it (only `--tune-compare K` does, after its timed region).        */
int b4d_plan_tune(b4d_plan* plan, const float* frames, int batch, float* psd, float psd_scale, float* autocorr,
                  unsigned flags, int candidates, float* best_ms, float* worst_ms, void* stream);

/* signal/corr.py:169-253 xcorr2d -- fftshift(ifft2(fft2(a) * conj(fft2(b)))), real part,
 * float32, scaled 1/(nx*ny) like ifft2.  B4D_REMOVE_MEAN zeroes the DC bin of the cross
 * spectrum (= both means removed), B4D_NORM_PEAK divides by max|corr|.                   */
int b4d_xcorr2d(b4d_plan* plan, const float* a, const float* b, int batch, float* corr, unsigned flags,
                void* stream);

/* signal/tracking.py:191-297 phase_correlation (backend="internal") for `npairs` (image,
 * template) pairs.  images: (nimg, ny, nx) raw frames.  Templates are ROIs of the frames in
 * tpl_src (ntplsrc, ny, nx): template k = frame tpl_frame[k], rows [roi[4k], roi[4k+1]),
 * columns [roi[4k+2], roi[4k+3]).  Pair i correlates image pair_img[i] with template
 * pair_tpl[i].  Every distinct image and template is transformed ONCE.  z-scoring
 * (tracking.py:308-311), zero-embedding (geometry/roi.py:175-222), whitening, |ifft2|,
 * first-occurrence arg-max, peak, exact median for the SNR (tracking.py:314-321) and the 3x3
 * Taylor step (324-375, including its swapped corrections) all run on the device.
 * The index arrays (tpl_frame, tpl_roi, pair_img, pair_tpl) are HOST pointers; out
 * (npairs, 4) float64 rows {dy, dx, peak, snr} and peak_ij (npairs, 2) int32 (nullable) are
 * DEVICE pointers.                                                                        */
int b4d_phase_correlation(b4d_plan* plan, const float* images, int nimg, const float* tpl_src, int ntplsrc,
                          const int32_t* tpl_frame, const int32_t* tpl_roi, int ntpl, const int32_t* pair_img,
                          const int32_t* pair_tpl, int npairs, int subpixel, double eps, double* out,
                          int32_t* peak_ij, void* stream);

/* signal/tracking.py:81-188 template_matching: zero-mean normalised cross-correlation of z-scored templates with
 * full frames over the "valid" window positions (the arithmetic of cv2.matchTemplate(TM_CCOEFF_NORMED) /
 * skimage.feature.match_template(pad_input=False), which the reference imports), first-occurrence arg-max, peak,
 * snr = |peak| / (median |ncc| + eps), 3x3 Taylor step, centre-to-centre shift against each template's ROI.
 * Same operands as b4d_phase_correlation, plus img_h x img_w: the extent of the images inside the plan's (ny, nx)
 * power-of-two canvas (0 = the whole canvas; frames of other sizes are zero-padded by the caller, which is exact for
 * the "valid" correlation).  zscore_image != 0: the image is z-scored as a whole ("opencv" path,
 * tracking.py:157); 0: raw float32 image ("skimage" path, tracking.py:166).  out: DEVICE (npairs, 4) float64
 * {dy, dx, peak, snr}; peak_ij: DEVICE (npairs, 2) int32 arg-max in the (ny-h+1, nx-w+1) map, or null.            */
int b4d_template_match(b4d_plan* plan, const float* images, int nimg, const float* tpl_src, int ntplsrc,
                       const int32_t* tpl_frame, const int32_t* tpl_roi, int ntpl, const int32_t* pair_img,
                       const int32_t* pair_tpl, int npairs, int img_h, int img_w, int zscore_image, int subpixel, double eps,
                       double* out, int32_t* peak_ij, void* stream);

/* Temporal per-pixel statistics (io/rw.py:129-132 for the mean).
 * accumulate: sum_x += sum_t x, sum_xx += sum_t x^2 over `nframes` frames of npix pixels
 * (float64 accumulators, caller zero-initialises; any npix / alignment).  finalize: mean, var (ddof 0),
 * contrast = sqrt(var)/mean as float32 maps from the (all-reduced) sums.                */
int b4d_temporal_accumulate(const float* frames, int nframes, size_t npix, double* sum_x, double* sum_xx,
                            void* stream);
int b4d_temporal_finalize(const double* sum_x, const double* sum_xx, double count, size_t npix, float* mean,
                          float* var, float* contrast, void* stream);
/* accumulate on pixels [pix0, pix0 + npix) of frames that are frame_stride pixels apart (row chunks of an image, so that
 * the all-reduce of finished rows overlaps the accumulation of the rest); sum_x / sum_xx point at the
 * accumulators of pixel pix0.  Any pixel count and alignment (odd frame sizes take a dword kernel). */
int b4d_temporal_accumulate_range(const float* frames, int nframes, size_t frame_stride, size_t pix0, size_t npix,
                                  double* sum_x, double* sum_xx, void* stream);
/* finalize with the frame count read from DEVICE memory: the count travels in the same all-reduced float64 buffer as the
 * sums, so the host never waits for it. */
int b4d_temporal_finalize_dev(const double* sum_x, const double* sum_xx, const double* count_dev, size_t npix, float* mean,
                              float* var, float* contrast, void* stream);

/* metrics/statistics.py:17-125 distribution_moments + speckles.py:640-645 visibility inputs:
 * per-frame finite-only power sums in float64.
 * out: (batch, 8) float64 {n_finite, sum, sum2(centered), sum3(centered), sum4(centered), n_zero, n_sat, min}... see DESIGN.md */
int b4d_moments(const float* frames, int batch, size_t npix, double eps, double saturation, double* out,
                void* stream);

/* metrics/sharpness.py:405-530 tenengrad + laplacian_variance: scipy.ndimage sobel/laplace
 * with mode="reflect", fused with their reductions.
 * out: (batch, 4) float64 {mean(gx^2), mean(gy^2), mean(lap), mean(lap^2)} over finite pixels. */
int b4d_sobel_laplace_stats(const float* frames, int batch, int ny, int nx, double* out, void* stream);

/* preprocessing/normalize.py:12-145 flat_field_correction, float32 arithmetic in the reference's order:
 *   b4d_stack_mean_f32   :86-93  mean of a (frames, npix) flat / dark stack along axis 0 (NumPy's float32 reduction
 *                                order: sequential adds in frame order, one division)
 *   b4d_flat_den         :107-113 den = F - D (D = 0 when dark is null); mask_bad != 0 writes NaN where den <= eps
 *                                (input of the median / mean selections, which skip NaN)
 *   b4d_flat_field       :115-132 out = ((I - D) / (F - D)) * scale, 0 where F - D <= eps; flat null: I - D
 *   b4d_repair_pixels    :134-140 replaces the listed pixels (idx: DEVICE int64 linear indices into one frame) by the
 *                                3x3 median (scipy "reflect") of the frame as it was on entry.  Synchronises the stream. */
int b4d_stack_mean_f32(const float* stack, int frames, size_t npix, float* out, void* stream);
int b4d_flat_den(const float* flat, const float* dark, size_t npix, float eps, int mask_bad, float* den, void* stream);
int b4d_flat_field(const float* frames, int batch, size_t npix, const float* flat, const float* dark, float eps, float scale,
                   int apply_scale, float* out, void* stream);
int b4d_repair_pixels(float* frames, int batch, int ny, int nx, const long long* idx, int nbad, void* stream);

/* images.astype(np.float32) on the device for raw detector words staged through pinned memory (barc4dip_amd/ingest.py):
 * dtype 0 u8, 1 u16, 2 i16, 3 i32, 4 u32, 5 f32, 6 f64.  src / dst: DEVICE, n elements. */
int b4d_to_f32(const void* src, int dtype, size_t n, float* dst, void* stream);

/* metrics/sharpness.py:752-861 eigenvalues (STA2): J = (x - mean(x)) / ||x||_2, eig_i = s_i(J)^2 / (M N - 1).
 * The reference takes every singular value from LAPACK and uses the first k (default 5); this returns the leading
 * nout (<= 8) of them, descending, from the Gram matrix of the smaller side (MFMA) and a 32-vector block subspace
 * iteration with float64 Cholesky-QR / Rayleigh-Ritz; frames with min(ny, nx) < 64 take every eigenvalue of the Gram
 * matrix from a parallel cyclic Jacobi in float64.
 * frames: DEVICE (batch, ny, nx) float32.  out: HOST (batch, nout) float64; NaN rows for frames holding non-finite
 * pixels or no energy.  Synchronises the stream.                                                                  */
int b4d_sta2_eigenvalues(const float* frames, int batch, int ny, int nx, double* out, int nout, void* stream);

/* utils/range.py:44-54 percentile_minmax_range / np.nanpercentile (linear interpolation): exact selection
 * of the two bracketing order statistics of the non-NaN pixels of every frame.  q: HOST array of nq (<= 16)
 * percentiles in [0, 100].  out: DEVICE (batch, nq, 4) float64 {x_lo, x_hi, fraction, n_valid}; the caller
 * finishes x_lo + (x_hi - x_lo) * fraction in float64.  Synchronises the stream.                          */
int b4d_percentiles(const float* frames, int batch, size_t npix, const double* q, int nq, double* out, void* stream);

/* maths/radial.py:101-169 radial_mean_interpolated: nr x ntheta polar samples, bilinear interpolation on the
 * pixel-centre grid, zero outside, mean over theta.  out: DEVICE (batch, nr) float64.                      */
int b4d_radial_profile(const float* maps, int batch, int ny, int nx, int nr, int ntheta, double r_max, double* out,
                       void* stream);

/* metrics/speckles.py:669-817 bandwidth + metrics/sharpness.py:536-629 spectral_entropy from a shifted PSD
 * map (DC bin treated as zero).  out: DEVICE (batch, 8) float64 {S_disc, sum FR^2 P, sum FX^2 P, sum FY^2 P,
 * sum P^2 (all four over the inscribed frequency disc), S_all, sum P ln P (all bins), f95 (square maps)}.  */
int b4d_psd_stats(const float* psd, int batch, int ny, int nx, double* out, void* stream);

/* preprocessing/filters.py:17-289 deconvolve_psf, method="wiener" (BASELINE.json config 5).
 * create: frame shape (h, w), PSF (ky, kx odd; HOST pointer, row-major float32, filters.py:217-230), Wiener-Hunt
 *   regularisation `balance` (skimage.restoration.wiener, Laplacian regulariser).  The padded size (h + 2*(ky/2),
 *   w + 2*(kx/2)) may be any integer whose odd part is <= 4200 (4096 + 8 = 4104 = 8 * 513 for sigma 1.5).
 * apply: per frame reflect-pad, divide by max|.|, filter in the Fourier domain of the padded size, clip to [-1, 1]
 *   (if clip), rescale, crop -> out (batch, h, w) float32.  Ordering is that of `stream`: a call with batch > 1 runs
 *   alternate frames on two plan-owned streams that wait for the work already queued on `stream` and that `stream`
 *   waits for before anything queued after the call (no host synchronisation).                                  */
typedef struct b4d_wiener b4d_wiener;
int b4d_wiener_create(int h, int w, const float* psf, int ky, int kx, float balance, b4d_wiener** out);
int b4d_wiener_apply(b4d_wiener* plan, const float* frames, int batch, float* out, int clip, void* stream);
int b4d_wiener_destroy(b4d_wiener* plan);

/* preprocessing/filters.py:270-277 method="rl": Richardson-Lucy deconvolution as published for
 * skimage.restoration.richardson_lucy (parity unpinned) with the reference's reflect padding by half the kernel,
 * normalisation by max|frame| and crop (filters.py:252-261, 287-289).  frames/out: DEVICE (batch, h, w) float32;
 * psf: HOST (ky, kx) float32, odd sides <= 33; filter_epsilon <= 0: none.  Synchronises the stream.            */
int b4d_richardson_lucy(const float* frames, int batch, int h, int w, const float* psf, int ky, int kx, int num_iter,
                        float filter_epsilon, int clip, float* out, void* stream);

/* preprocessing/filters.py:278-286 method="uw": ONE Gibbs sweep of skimage.restoration.unsupervised_wiener (published
 * algorithm, parity unpinned and stochastic: the reference passes no rng) over the unitary half-plane spectrum.  All pointers
 * DEVICE.  y, tf, x_sample (optional), postmean: (ny, nxh) complex64 (rfft2 layout, nxh = nx/2 + 1); areg2: (ny, nxh) float32
 * = |Laplacian transfer function|^2; r1 / r2: (ny, nxh) float32 standard normals, or both null: generated on the device
 * (Philox-4x32-10, key `seed`, counter = element and sweep).  Does
 *   x = gn conj(tf) / (gn |tf|^2 + gx areg2) * y + sqrt(0.5 / (..)) (r1 + i r2);  postmean += x for sweep > burnin
 * and leaves in sums4 (4 doubles): ||y - x tf||^2, ||x L||^2 (half-plane weights as the library's image_quad_norm),
 * sum |postmean/(sweep-burnin) - previous/(sweep-burnin-1)| and sum |postmean| (0 before they are defined).  The Gamma
 * draws of the two precisions and the loop belong to the caller.  Asynchronous on `stream`.                       */
int b4d_uw_step(const void* y, const void* tf, const float* areg2, void* x_sample, void* postmean, const float* r1,
                const float* r2, unsigned long long seed, int sweep, int burnin, float gn, float gx, int ny, int nxh,
                double* sums4, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* B4D_H */
