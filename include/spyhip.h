/*
 * spyhip.h - C ABI of the MI355X (gfx950) spectral-estimation / cross-spectral
 * connectivity hot path.
 *
 * This is the drop-in boundary underneath Syncopy's `computeFunction` plug-in
 * surface (reference: syncopy/shared/computational_routine.py:51-1108 and the
 * cF contract in doc/source/developer/compute_kernels.rst:63-88).  The
 * reference has no native code; every entry point below names the reference
 * *Python* function whose arithmetic it replaces.  Python binds this header
 * through ctypes (syncopy_amd/backend.py); INTEGRATION.md shows the stub a
 * Syncopy maintainer would add.
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error; spyhip_last_error()
 *    returns a thread-local message for the last failure;
 *  - pointers suffixed `_d` are DEVICE pointers (HBM), all others are host
 *    pointers that are consumed before the call returns;
 *  - all work is enqueued on the context's HIP stream (spyhip_ctx_set_stream),
 *    nothing synchronises unless stated; caller owns every buffer;
 *  - complex64 = interleaved (re, im) float pairs, complex128 = double pairs;
 *  - trial data live in ONE (rows x ld) float32 matrix, channel fastest, as
 *    AnalogData does (syncopy/datatype/continuous_data.py:405); a "segment" is
 *    `nsig` consecutive rows starting at `seg_start[b]` of which only rows in
 *    [seg_lo[b], seg_hi[b]) are read (others count as 0.0f - this is the zero
 *    extension of stft.py:101-117); a trial is the segment
 *    start=lo=sampleinfo[t,0], hi=sampleinfo[t,1].
 */
#ifndef SPYHIP_H
#define SPYHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct spyhip_ctx spyhip_ctx;
typedef struct spyhip_fft_plan spyhip_fft_plan;
typedef struct spyhip_cwt_plan spyhip_cwt_plan;
typedef struct spyhip_queue spyhip_queue;

/* output conversions = syncopy/shared/const_def.py:25-37 `spectralConversions` */
enum spyhip_output {
    SPYHIP_OUT_POW = 0,
    SPYHIP_OUT_ABS = 1,
    SPYHIP_OUT_FOURIER = 2, /* complex64 ("fourier" / "complex") */
    SPYHIP_OUT_REAL = 3,
    SPYHIP_OUT_IMAG = 4,
    SPYHIP_OUT_ANGLE = 5,
    SPYHIP_OUT_ABSREAL = 6,
    SPYHIP_OUT_ABSIMAG = 7
};

/* polynomial removal before tapering = scipy.signal.detrend call sites
 * specest/compRoutines.py:169-172, connectivity/ST_compRoutines.py:405-409,
 * per segment in specest/stft.py:131-132 */
enum spyhip_detrend { SPYHIP_DETREND_NONE = -1, SPYHIP_DETREND_CONSTANT = 0, SPYHIP_DETREND_LINEAR = 1 };

/* ---- context ------------------------------------------------------------ */
int spyhip_version(void);
const char* spyhip_last_error(void);
int spyhip_ctx_create(int device, spyhip_ctx** ctx);
int spyhip_ctx_destroy(spyhip_ctx* ctx);
/* `stream` is a hipStream_t (NULL = the null stream) */
/* Give back the device memory a context keeps between calls (split-launch scratch, the work arrays of spyhip_granger). */
int spyhip_ctx_trim(spyhip_ctx* ctx);
int spyhip_ctx_set_stream(spyhip_ctx* ctx, void* stream);
int spyhip_ctx_synchronize(spyhip_ctx* ctx);

/* ---- device memory ---------------------------------------------------------
 * The library can be driven from a host that has nothing but this header (NumPy + ctypes, C, ...): it allocates
 * HBM, moves data and synchronises by itself.  Hosts that already own device memory (PyTorch tensors) pass their
 * pointers instead - the two can be mixed freely.  Copies are enqueued on the context's stream and are complete
 * when the call returns (the stream is synchronised). */
int spyhip_alloc(spyhip_ctx* ctx, size_t bytes, void** ptr_d);
int spyhip_free(spyhip_ctx* ctx, void* ptr_d);
int spyhip_memset(spyhip_ctx* ctx, void* ptr_d, int value, size_t bytes);            /* asynchronous */
int spyhip_upload(spyhip_ctx* ctx, void* dst_d, const void* src, size_t bytes);
int spyhip_download(spyhip_ctx* ctx, void* dst, const void* src_d, size_t bytes);

/* ---- the in-HBM trial queue -------------------------------------------------
 * Replaces the per-trial HDF5 slab reads of ComputationalRoutine.compute_sequential
 * (shared/computational_routine.py:1001-1007) and the trial bookkeeping of AnalogData
 * (datatype/continuous_data.py:255,405; sampleinfo: datatype/base_data.py:993): the whole (nrows x nchan) float32
 * matrix goes to HBM once, together with the trials' row ranges.  Trial t = rows
 * [sampleinfo[2t], sampleinfo[2t+1]) - any order, overlaps and repeats allowed, exactly as given (integer work).
 * spyhip_queue_segments hands out the device arrays spyhip_fft_exec / spyhip_cwt_exec take as
 * seg_start_d (= seg_lo_d) and seg_hi_d. */
int spyhip_queue_upload(spyhip_ctx* ctx, const float* data, int64_t nrows, int nchan, const int64_t* sampleinfo,
                        int ntrials, spyhip_queue** queue);
int spyhip_queue_destroy(spyhip_queue* queue);
const float* spyhip_queue_data(const spyhip_queue* queue);            /* device pointer, ld = nchan */
int spyhip_queue_segments(const spyhip_queue* queue, const int64_t** start_d, const int64_t** stop_d, int* ntrials);

/* ---- C1: sums over ranks (RCCL) ---------------------------------------------
 * Replaces the mutex-guarded `+=` of the reference's parallel trial map (shared/kwarg_decorators.py:723-735,
 * shared/computational_routine.py:939-942): one process per GPU, each with the accumulator of its trial shard.
 * Bootstrap: ONE rank calls spyhip_comm_unique_id, the host distributes the 128 bytes by any means it has (file,
 * environment, MPI, a torch.distributed store ...), every rank calls spyhip_comm_init with the same id.  RCCL
 * (librccl.so) is loaded on first use; single-GPU users never need it.
 * spyhip_allreduce_csd: acc_d complex64 (nfreq, nchan, nchan) raw accumulator of spyhip_csd_accumulate; only its
 *   lower triangle carries data, so that is what travels (packed into library scratch, summed over all ranks,
 *   unpacked): 0.54 GB instead of 1.07 GB at 2049 x 256 x 256.
 * spyhip_allreduce: in-place sum of n float32 (dtype 0) or float64 (dtype 1) values (phasor sums of K7, the
 *   jackknife sums, trial sums of averaged spectra; complex arrays as interleaved reals).
 * All are enqueued on the context's stream.  With a communicator of one rank they run all the same (identity). */
#define SPYHIP_UNIQUE_ID_BYTES 128
int spyhip_comm_unique_id(void* id_out);
int spyhip_comm_init(spyhip_ctx* ctx, const void* id, int rank, int nranks);
int spyhip_comm_destroy(spyhip_ctx* ctx);
int spyhip_comm_info(const spyhip_ctx* ctx, int* rank, int* nranks);     /* -1 if there is no communicator */
int spyhip_allreduce_csd(spyhip_ctx* ctx, void* acc_d, int nfreq, int nchan);
int spyhip_allreduce(spyhip_ctx* ctx, void* buf_d, int64_t n, int dtype);

/* ---- K1/K2: (multi-)tapered FFT of segments ------------------------------
 * Replaces mtmfft (specest/mtmfft.py:16-129) + the tail of mtmfft_cF
 * (specest/compRoutines.py:169-189), and, with nsig == nfft == nperseg and
 * one segment per frame, stft/mtmconvol (specest/stft.py:16-159,
 * specest/mtmconvol.py:136-150).
 *
 *   X[b,k,f,c] = scale * sum_{n<nsig} w[k,n] * x_b[n,c] * exp(-2 pi i f n / nfft)
 *
 * tapers : K x nsig float64, already normalised as _norm_taper
 *          (specest/_norm_spec.py:27-46); scale as _norm_spec (:10-24).
 * detrend: enum spyhip_detrend, applied per segment to the (zero-extended)
 *          nsig samples; demean_taper: subtract the per-channel mean again
 *          after tapering (mtmfft.py:115-116).
 * freq_idx: nfsel indices into the nfft/2+1 rfft bins (NULL = all bins);
 *          duplicates must already be squashed (shared/tools.py:334-336).
 * output  : enum spyhip_output; keeptapers=0 averages the converted values
 *          over tapers (compRoutines.py:188-189).
 * Result  : (B, Kout, nfsel, nchan) float32 or complex64, Kout = keeptapers ? K : 1.
 */
int spyhip_fft_plan_create(spyhip_ctx* ctx, int nsig, int nfft, int nchan, int ntaper,
                           const double* tapers, double scale, int detrend, int demean_taper,
                           const int32_t* freq_idx, int nfsel, int output, int keeptapers,
                           spyhip_fft_plan** plan);
int spyhip_fft_plan_destroy(spyhip_fft_plan* plan);
/* data_d: float32 (rows x ld); chan_idx_d: nchan column indices or NULL
 * (=0..nchan-1); seg_*_d: B int64 each; out_d: see above. */
int spyhip_fft_exec(spyhip_fft_plan* plan, const float* data_d, int64_t ld,
                    const int32_t* chan_idx_d, const int64_t* seg_start_d, const int64_t* seg_lo_d,
                    const int64_t* seg_hi_d, int nseg, void* out_d);
/* Internal hand-over layout between spyhip_fft_exec and spyhip_csd_accumulate_blocked (the coherence path
 * never shows the per-trial spectra to the host): with on != 0 a FOURIER / keeptapers=1 plan writes
 * (nseg*ntaper, ceil(nchan/4), nfsel, 4) complex64 - the four channels of a quad contiguous in frequency -
 * so that every wave stores whole 2-KiB runs.  Channels beyond nchan in the last quad are written as 0.
 * Returns -3 for plans that cannot use it (other outputs, nfft not a power of two in 256..8192). */
int spyhip_fft_plan_set_blocked(spyhip_fft_plan* plan, int on);
/* Precision of the transform.  reference = 0 (default): taper product and FFT in float32 - error ~1e-7 of a
 * channel's largest bin, inside the parity criterion |a-b| <= 1e-5 |b| + 1e-6 max|b|.  reference = 1: float64 taper
 * product and float64 FFT, the spectrum rounded to complex64 and scaled in float32 exactly where
 * specest/mtmfft.py:104-127 does it - every bin to ~1e-7 of ITSELF (pure rtol 1e-5 also 60 dB below the peak).  Cost:
 * ~2x float32 where a compile-time radix schedule exists (mtmfft_dec64_kernel.h: nfft = 256 ... 16384 powers of two, 200,
 * 500, 1000, 2000, 2500, 4000, 5000, 10000); every other nfft up to 2^20 runs generic Stockham passes over work arrays in
 * LDS (nfft <= 5120) or global memory (4-10x), in Bluestein's chirp-z form when nfft has a prime factor above 61;
 * standard layout only; -3 beyond 2^20 points. */
int spyhip_fft_plan_set_precision(spyhip_fft_plan* plan, int reference);
/* Constant detrending (detrend = SPYHIP_DETREND_CONSTANT) with the per-channel mean taken EXACTLY as the reference
 * takes it for whole trials: scipy.signal.detrend on the float32 (time x channel) array is data - np.mean(data, 0),
 * and NumPy sums the slow axis sequentially in ONE float32 accumulator per channel (then one float32 division).
 * For channels riding on an offset that rounding sequence is what the bins next to DC consist of.  on = 1: a
 * pre-pass reproduces it literally (one thread per channel walks the rows in order); on = 0 (default): float64
 * block sums inside the transform kernel - the better match for per-segment detrending of sliding windows, which
 * the reference does in float64 (specest/stft.py:112-132); on = 2: the same, and the plan is told that the reference
 * holds these segments as FLOAT64 arrays (zero-extended / padded windows): the reference-precision kernels
 * (spyhip_fft_plan_set_precision) then subtract the trend in float64 instead of rounding the samples to float32. */
int spyhip_fft_plan_set_reference_mean(spyhip_fft_plan* plan, int on);
/* Range of the spectra for spyhip_csd_accumulate_split: with absmax_d != NULL (nchan floats on the device, zeroed by the
 * caller) every spyhip_fft_exec of a FOURIER / keeptapers=1 plan raises absmax_d[c] to a BOUND of every |re|, |im| it
 * writes for channel c: max over tapers of ||w scale||_2 times the 2-norm of the detrended segment (Cauchy-Schwarz; one sum
 * of squares per segment from samples that are in registers anyway - a maximum over the values written costs the
 * transform kernel 6 %).  The bound sits sqrt(nsig) / (crest factor) above the largest bin of noise-like data (3-4 bits of
 * the ~18 the half-precision kernel has) and is tight for offset- or line-dominated channels, where the bits matter.
 * Float32 transforms of a power-of-two length 256 ... 8192 form it inside the transform kernel; every other length and the
 * float64 transforms (spyhip_fft_plan_set_precision) take one pass over the segments ahead of the transform.  Returns -3
 * (and stores nothing) for plans that do not write complex all-taper spectra in the standard layout; NULL switches it off. */
int spyhip_fft_plan_set_absmax(spyhip_fft_plan* plan, float* absmax_d);
/* name of the dominant kernel a plan launches (for rocprof matching) */
const char* spyhip_fft_plan_kernel_name(const spyhip_fft_plan* plan);

/* ---- K4: cross-spectral density accumulation (MFMA) -----------------------
 * Replaces the outer product + taper mean of csd (connectivity/csd.py:94-115),
 * spectral_dyadic_product_cF (connectivity/ST_compRoutines.py:30-117) and the
 * trial sum of ComputationalRoutine.compute_sequential
 * (shared/computational_routine.py:1022-1032):
 *
 *   acc[f,i,j] += sum_{r<nrows} X[r,f,i] * conj(X[r,f,j])        (i >= j)
 *
 * spec_d: complex64 (nrows, nfreq, nchan), nrows = trials x tapers (the
 * layout spyhip_fft_exec writes with output=FOURIER, keeptapers=1);
 * acc_d: complex64 (nfreq, nchan, nchan); only the lower triangle (incl.
 * diagonal, 32x32 tile granularity) is maintained until spyhip_csd_finalize. */
int spyhip_csd_accumulate(spyhip_ctx* ctx, const void* spec_d, int64_t nrows, int nfreq, int nchan,
                          void* acc_d);
/* Arithmetic of the accumulation.  Default (0): the 3-multiplication complex product where a kernel is built for the
 * channel count: re = P1 + P2, im = P3 - P1 + P2 from three fp32 row sums.  Its imaginary part carries an ABSOLUTE
 * error of ~6e-8 * sqrt(nrows) * |Re acc| (three independently rounded sums are subtracted): within rtol 1e-5 of
 * the complex value and of abs / pow / real outputs, but not of the imaginary part or the phase of strongly
 * coherent channel pairs near zero lag.  on = 1: the 4-multiplication kernels everywhere, whose imaginary part is
 * summed directly like the reference's complex64 products (connectivity/csd.py:98-102) - what the front ends select
 * for output = "imag" / "angle".  Per context; costs ~25 % of K4's throughput at 256 channels. */
int spyhip_csd_set_phase_exact(spyhip_ctx* ctx, int on);
/* The same accumulation on the HALF-PRECISION matrix cores (K4h, csrc/csdh_kernel.h) for nchan = 256: every float32
 * operand, scaled by a power of two per channel, is split once into an fp16 pair hi + lo (22 significant bits) and a
 * real product is hi hi' + hi lo' + lo hi' with float32 accumulation - float32-class products at 5.3 x less matrix time
 * than the float32 instructions, with the plain 4-multiplication complex product (the imaginary part is summed directly:
 * this path also serves spyhip_csd_set_phase_exact contexts).  absmax_d: 256 floats on the device, per channel an upper
 * bound of |re| and |im| over the spectra of this call - what spyhip_fft_plan_set_absmax makes spyhip_fft_exec deliver
 * for free - or NULL: the library takes one extra pass over the spectra.  A frequency where a channel's rms sits more
 * than ~2^18 below that bound (or that holds Inf / NaN) is left to the float32 kernels by the half-precision kernel
 * itself (checked on the diagonal it accumulated, before anything is added): results never depend on the data's dynamic
 * range, only the speed does.  Other channel counts, and SPYHIP_CSD_F32=1 in the environment: identical to
 * spyhip_csd_accumulate.  Same reference lines (connectivity/csd.py:94-102). */
int spyhip_csd_accumulate_split(spyhip_ctx* ctx, const void* spec_d, int64_t nrows, int nfreq, int nchan,
                                void* acc_d, const float* absmax_d);
/* The same update for the frequencies [f0, f0 + nf) only (spec_d and acc_d still point at frequency 0; nchan = 256 and
 * absmax_d are required): a caller that launches range by range can normalise and ship the results of range r while range
 * r + 1 is being accumulated - the coherence pipeline of the front end hides three quarters of its 0.54 GB host copy this
 * way.  The ranges of one accumulation may come in any order but must not overlap; together they equal one full call; a
 * range that reaches into the last partial round of workgroups (nfreq % CUs frequencies at the end) must end at nfreq.
 * Reference lines: connectivity/csd.py:94-102 (products), shared/computational_routine.py:1022-1036 (a result the caller
 * can read). */
int spyhip_csd_accumulate_split_range(spyhip_ctx* ctx, const void* spec_d, int64_t nrows, int nfreq, int nchan,
                                      void* acc_d, const float* absmax_d, int f0, int nf);
/* number of frequencies the last spyhip_csd_accumulate_split call of this context handed to the float32 kernels
 * (synchronises the stream; for tests and benchmarks) */
int spyhip_csd_split_fallbacks(spyhip_ctx* ctx, int* count);
/* same accumulation from spectra in the channel-blocked layout of spyhip_fft_plan_set_blocked:
 * spec_d = (nrows, ceil(nchan/4), nfreq, 4) complex64.  Bit-identical results. */
int spyhip_csd_accumulate_blocked(spyhip_ctx* ctx, const void* spec_d, int64_t nrows, int nfreq, int nchan,
                                  void* acc_d);
/* Lower triangle (i >= j) of the accumulator <-> packed (nfreq, nchan(nchan+1)/2) complex64: what the multi-GPU
 * path all-reduces between spyhip_csd_accumulate and spyhip_csd_finalize (the mutex-guarded `+=` of
 * shared/kwarg_decorators.py:723-735 ships 0.54 GB instead of 1.07 GB at 256 channels).  Unpack writes only the lower
 * triangle of acc_d. */
int spyhip_csd_tril_pack(spyhip_ctx* ctx, const void* acc_d, int nfreq, int nchan, void* packed_d);
int spyhip_csd_tril_unpack(spyhip_ctx* ctx, const void* packed_d, int nfreq, int nchan, void* acc_d);
/* acc[f,i,j] *= scale on the lower triangle and acc[f,j,i] = conj(acc[f,i,j]):
 * scale = 1/(ntaper*ntrials) turns the sum into the taper- and trial-mean. */
int spyhip_csd_finalize(spyhip_ctx* ctx, void* acc_d, int nfreq, int nchan, double scale);

/* ---- K5: coherence normalisation ------------------------------------------
 * Replaces normalize_csd (connectivity/csd.py:118-172):
 *   out[f,i,j] = conv( csd[f,i,j] / sqrt(csd[f,i,i]*csd[f,j,j]) )
 * output: POW/ABS/FOURIER/REAL/IMAG/ANGLE; out_d float32 or complex64
 * (nfreq, nchan, nchan); csd_d must be a full Hermitian array. */
int spyhip_coh_normalize(spyhip_ctx* ctx, const void* csd_d, int nfreq, int nchan, int output,
                         void* out_d);

/* K5 fused for the coherence pipeline: out = conv( (acc*scale)[f,i,j] / sqrt((acc*scale)[f,i,i] (acc*scale)[f,j,j]) ) for the
 * whole Hermitian (nfreq, nchan, nchan) output, read from the RAW lower-triangle accumulator of
 * spyhip_csd_accumulate (acc_d is not modified).  Equals spyhip_csd_finalize + spyhip_coh_normalize bit for bit
 * at a third of the HBM traffic. */
int spyhip_coh_from_accumulator(spyhip_ctx* ctx, const void* acc_d, int nfreq, int nchan, double scale, int output,
                                void* out_d);

/* ---- K3s: superlets ----------------------------------------------------------
 * spyhip_cwt_plan_create_sl: a CWT plan (same exec / destroy as above) whose kernels are the superlet formulation
 *   MorletSL (specest/superlet.py:268-292) sampled as cwtSL does (:311-375): support 10*s*cycles/dt samples, norm
 *   sqrt(dt)/(4 pi), k_sd standard deviations of the Gaussian envelope (the reference uses 5).
 * spyhip_slt_combine: one factor of the geometric mean of multiplicativeSLT / FASLT (superlet.py:97-211):
 *   acc[r, s0+q, c] = (init ? 1 : acc[r, s0+q, c]) * spec[r, q, c] ^ expo[q]     principal branch, 0^e = 0 (e > 0)
 *   acc_d complex64 (nrows, nscales, nchan); spec_d complex64 (nrows, nsub, nchan) = FOURIER output of one order's
 *   plan over the scales [s0, s0+nsub); expo = nsub host doubles (0 leaves the element as it is); modulus_only = 1
 *   folds |spec|^expo instead (all that the POW / ABS outputs need: no phase arithmetic); modulus_only = 2: acc_d and
 *   spec_d are float32 arrays of moduli (ABS output of the plan), half the traffic; | 4: the product is squared on
 *   store (last factor of a POW output).
 * spyhip_spec_convert: spectralConversions (shared/const_def.py:25-33) of n complex64 values to a real output kind. */
int spyhip_cwt_plan_create_sl(spyhip_ctx* ctx, int nsig, int nchan, int nscales, const double* scales, double dt,
                              double cycles, double k_sd, int detrend, int output, const int32_t* tpos, int ntime_out,
                              spyhip_cwt_plan** out);
int spyhip_slt_combine(spyhip_ctx* ctx, void* acc_d, const void* spec_d, int64_t nrows, int nscales, int nsub, int s0,
                       int nchan, const double* expo, int init, int modulus_only);
int spyhip_spec_convert(spyhip_ctx* ctx, const void* in_d, int64_t n, int output, void* out_d);

/* ---- K7: pairwise phase consistency -----------------------------------------
 * Replaces ppc_column_cF (connectivity/ST_compRoutines.py:159-233) and the loop over all trial pairs with its
 * weighted average (connectivity/connectivity_analysis.py:624-663):
 *   ppc[f,i,j] = 2/(T(T-1)) sum_{a<b} cos(arg(S_a[f,i,j] conj(S_b[f,i,j])))  =  (|sum_t u_t|^2 - T) / (T(T-1)),
 *   u_t = S_t/|S_t| (1 where S_t = 0, as np.angle(0) = 0),  S_t = taper mean of X conj(X)^T of trial t.
 * spyhip_ppc_accumulate: acc[f,i,j] += sum_t u_t straight from the tapered spectra of spyhip_fft_exec
 *   (spec_d complex64 (ntrials*ntaper, nfreq, nchan), the ntaper rows of a trial adjacent); acc_d complex64
 *   (nfreq, nchan, nchan), maintained on the lower triangle (32x32 tile granularity).
 * spyhip_ppc_accumulate_csd: the same sum from single-trial cross spectra that exist already
 *   (csd_d complex64 (ntrials, nelem), acc_d complex64 (nelem): any block shape, e.g. channelcmb rectangles).
 * spyhip_ppc_finalize: out float32 (nfreq, ni, nj) from the accumulator and the total number of trials;
 *   lower_only = 1 for accumulators of spyhip_ppc_accumulate (the upper triangle is mirrored). */
int spyhip_ppc_accumulate(spyhip_ctx* ctx, const void* spec_d, int ntrials, int ntaper, int nfreq, int nchan,
                          void* acc_d);
int spyhip_ppc_accumulate_csd(spyhip_ctx* ctx, const void* csd_d, int ntrials, int64_t nelem, void* acc_d);
int spyhip_ppc_finalize(spyhip_ctx* ctx, const void* acc_d, int nfreq, int ni, int nj, int lower_only,
                        int64_t ntrials, void* out_d);

/* ---- K9: streaming jackknife of the coherence --------------------------------
 * Replaces, for method='coh' with jackknife=True, the T leave-one-out trial averages (statistics/jackknifing.py:14-108),
 * the AV stage on every replicate (connectivity_analysis.py:736-745) and the sums behind bias_var (:111-184):
 *   for every trial t of the batch:  S_t = taper mean of X conj(X)^T,  loo = (T*S - S_t)/(T-1)  (complex64),
 *   c_t = conv(loo_ij / sqrt(loo_ii loo_jj)),  d_t = c_t - direct;   sum_d += d_t,  sum_d2 += |d_t|^2   (float64)
 * spec_d complex64 (ntrials*ntaper, nfreq, nchan) as for spyhip_ppc_accumulate; csd_d complex64 (nfreq, nchan, nchan)
 * = the finalised trial average S; direct_d = spyhip_coh_normalize(S) in the same output kind (float32, or complex64
 * for FOURIER); sum_d float64 (nfreq, nchan, nchan) - or complex128 for FOURIER -, sum_d2 float64; T = ntrials_total.
 * bias = (T-1) sum_d / T,  var = (T-1) (sum_d2 - |sum_d|^2 / T). */
int spyhip_jack_coh_accumulate(spyhip_ctx* ctx, const void* spec_d, int ntrials, int ntaper, int nfreq, int nchan,
                               const void* csd_d, const void* direct_d, int output, int64_t ntrials_total,
                               void* sum_d, void* sum_d2);

/* ---- K8: cross-covariance / cross-correlation -------------------------------
 * Replaces cross_covariance_cF (connectivity/ST_compRoutines.py:466-584: one fftconvolve per channel pair and
 * trial, lags 0 .. N/2 divided by the overlap N - lag, norm=True: divided by the products of np.std), the trial
 * average, and normalize_ccov_cF (connectivity/AV_compRoutines.py:166-228).
 * The trial sum is taken on the cross spectra: feed spyhip_csd_accumulate with the spectra of the trials
 * zero-padded to nfft = spyhip_ccov_nfft(nsamples) points (spyhip_fft_exec, boxcar, FOURIER; 1 row per trial),
 * then ONE inverse transform per channel pair:
 *   out[l,a,b] = scale * R_ab(l) / (N - l)         a >= b,   R_ab(tau) = sum_n x_a[n] x_b[n - tau]
 *   out[l,a,b] = scale * R_ab(l + q) / (N - l)     a <  b,   q = 1 for even N, 0 for odd N (the reference's
 *                                                            reversed "same" crop), l = 0 .. ceil(N/2) - 1
 * acc_d: raw lower-triangle accumulator (nfft/2 + 1, nchan, nchan) complex64; out_d float32 (ceil(N/2), nchan,
 * nchan); scale = 1 / (ntrials * fft_scale^2) (the 1/nfft of the inverse transform is applied inside).  norm: 0 none; 1 divide by sqrt(out[0,a,a] out[0,b,b])
 * (normalize_ccov_cF); 2 divide by np.std(x_a) np.std(x_b) of the ONE trial in acc_d (cross_covariance_cF norm=True).
 * spyhip_ccov_nfft: transform length for trials of nsamples (power of two >= N + ceil(N/2), 1024 .. 8192), or -1. */
int spyhip_ccov_nfft(int nsamples);
int spyhip_ccov_from_accumulator(spyhip_ctx* ctx, const void* acc_d, int nfft, int nchan, int nsamples, double scale,
                                 int norm, void* out_d);
/* normalize_ccov_cF alone, in place on a cross-covariance cc_d float32 (nlag, nchan, nchan):
 * cc[l,a,b] /= sqrt(cc[0,a,a] cc[0,b,b]). */
int spyhip_ccov_normalize(spyhip_ctx* ctx, void* cc_d, int nlag, int nchan);

/* ---- K3: Morlet continuous wavelet transform ------------------------------
 * Replaces cwt_time (specest/wavelets/transform.py:88-108) with Morlet.time
 * (specest/wavelets/wavelets.py:27-86) and the tail of wavelet_cF
 * (specest/compRoutines.py:582-595): detrend the whole trial, full linear
 * convolution of the nsig pre-selected samples with the sampled complete Morlet
 * kernel of every scale (M = 10*s/dt taps, amplitude sqrt(dt)/(8 pi s)), cropped
 * like scipy.signal.fftconvolve(mode="same"), converted with `output`.
 * tpos (host, nsig entries or NULL): output slot of sample n or -1 (post-selection).
 * Result per segment: (ntime_out, 1, nscales, nchan) float32 / complex64. */
int spyhip_cwt_plan_create(spyhip_ctx* ctx, int nsig, int nchan, int nscales, const double* scales,
                           double dt, double w0, int detrend, int output, const int32_t* tpos,
                           int ntime_out, spyhip_cwt_plan** plan);
/* The same transform with another sampled kernel family (the convolution kernels do not care which):
 * family 0 Morlet(w0 = p0), 1 MorletSL(cycles = p0, k_sd = p1) (= spyhip_cwt_plan_create_sl), 2 Paul(m = p0),
 * 3 DOG(m = p0) - Ricker / Marr / Mexican_hat are DOG(2) (specest/wavelets/wavelets.py:140-223; freqanalysis.py:55). */
int spyhip_cwt_plan_create_family(spyhip_ctx* ctx, int nsig, int nchan, int nscales, const double* scales,
                                  double dt, int family, double p0, double p1, int detrend, int output,
                                  const int32_t* tpos, int ntime_out, spyhip_cwt_plan** plan);
int spyhip_cwt_plan_destroy(spyhip_cwt_plan* plan);
/* reference = 1: the transform as scipy.signal.fftconvolve computes it for cwt_time / cwtSL (specest/wavelets/
 * transform.py:88-108, specest/superlet.py:311-375): float64 FFT convolution of the detrended float32 trial with the
 * complex128 taps, rounded to complex64 where the reference stores it - one length-2^m >= nsig + taps - 1 convolution per
 * (segment, channel), generic Stockham passes over work arrays in global memory (~50x the float32 kernels: for
 * precision="reference" and the per-trial route, where the copies cost as much).  0 (default): the float32 overlap-save
 * kernels (~5e-7 of a trial's largest coefficient). */
int spyhip_cwt_plan_set_precision(spyhip_cwt_plan* plan, int reference);
/* on = 1 (default): scales whose kernel support fits 1024- / 2048-point blocks leave their transform kernel in the output's
 * own (segment, time, scale, channel) layout (cwt2d_kernel: 16 / 8 channels per workgroup, 64- / 32-byte runs); trial sums
 * (accumulate = 2) are read-modify-writes of tiles a workgroup owns.  0: every scale through the time-contiguous staging
 * buffer and the transposition pass (rounds 1-5; kept for A/B measurements and as the cross-check of the direct kernels).
 * Replaces the (nScales, N, C) array the reference writes once per trial: specest/wavelets/transform.py:88-108. */
int spyhip_cwt_plan_set_direct(spyhip_cwt_plan* plan, int on);
/* seg_start_d: row of sample 0 of each pre-selected signal; trial_lo_d/trial_hi_d: rows of the
 * whole trial (detrending range); accumulate: 0 = store, 1 = out_d[b] += result of segment b,
 * 2 = out_d[0] += sum over the nseg segments (trial averaging: one read-modify-write of the output per chunk
 * of segments instead of one per trial). */
int spyhip_cwt_exec(spyhip_cwt_plan* plan, const float* data_d, int64_t ld,
                    const int32_t* chan_idx_d, const int64_t* seg_start_d, const int64_t* trial_lo_d,
                    const int64_t* trial_hi_d, int nseg, void* out_d, int accumulate);

/* ---- K6: Wilson spectral factorisation + Granger-Geweke causality ---------
 * Replaces regularize_csd / wilson_sf (connectivity/wilson_sf.py:16-254) and
 * granger (connectivity/granger.py:10-79) as called from granger_cF
 * (connectivity/AV_compRoutines.py:293-412).  csd_d: complex64 (nfreq, C, C)
 * trial-averaged CSD.  granger_d: float32 (nfreq, C, C).  info (host, 4
 * doubles): converged, max rel. err, reg. factor, initial cond. num.
 * H_d / Sigma_d (complex128 (nfreq,C,C) / complex128 (C,C)) may be NULL.
 * Return -6: a Cholesky factorisation met a matrix that is not positive definite (np.linalg.cholesky's LinAlgError at
 * wilson_sf.py:76,144-151; the Python host mirror raises that type). */
int spyhip_granger(spyhip_ctx* ctx, const void* csd_d, int nfreq, int nchan, double rtol, int niter,
                   double cond_max, double eps_max, void* granger_d, void* H_d, void* Sigma_d,
                   double* info);

/* ---- K6 in steps, for frequency shards ---------------------------
 * One rank holds the bins [f_lo, f_lo + nf) of the nftot rfft bins of the trial-averaged CSD.  Regularisation,
 * Cholesky factor, inverse, products and the error are per frequency (local); the plus operator
 * (wilson_sf.py:154-184) works along the frequency axis: the host transposes g between "my frequencies x all entries"
 * and "all frequencies x my entries" around spyhip_wilson_plus (one all-to-all each way) and reduces three small
 * quantities over ranks (gamma_0: sum; condition number, error: max).  Host side:
 * syncopy_amd/connectivity/wilson_sharded.py; with one rank the sequence equals spyhip_granger.
 * Arrays are complex128 on the device unless stated; "work" arrays are scratch of the stated size.
 *   spyhip_wilson_cond   A = complex128(csd) + eps I (regularize_csd, wilson_sf.py:239-248); *cond_out = largest
 *                        2-norm condition number of the local bins.  work_d: 3 nf n^2.
 *   spyhip_wilson_init   U = Cholesky factor of A per bin (:76); gamma_part_d (n, n) = this shard's part of
 *                        gamma_0 = fft(CSD_full)[0] (:135-140), to be summed over ranks.
 *   spyhip_wilson_psi0   psi0 = chol(gamma_0)^T (:144-151) from the summed gamma_0 (overwritten), tiled into psi_d.
 *   spyhip_wilson_g      g = (psi^-1 U)(psi^-1 U)^H + I (:80-92).  work_d: 2 nf n^2.  Returns 1 (not an error) if
 *                        the block inverse met a tiny pivot: restart the factorisation with pivoted = 1.
 *   spyhip_wilson_plus   g+ and the halved zero-lag coefficients g0 for nent entries over all nftot frequencies:
 *                        g_d, gp_d (nftot, nent), g0_d (nent).
 *   spyhip_wilson_update psi <- psi (g+ + S), psi0 <- psi0 (g0 + S), S = triu(g0) - triu(g0)^H (:97-101);
 *                        *err_out = this shard's max |A - psi psi^H| / |A| (:103,190-194).  g0_d: all n^2 entries.
 *                        work_d: nf n^2.
 *   spyhip_wilson_finish Sigma = psi0 psi0^T, H = psi psi0^-1, Granger causality (granger.py:53-77) on the local
 *                        bins: granger_d float32 (nf, n, n); H_d / Sigma_d may be NULL.  work_d: nf n^2. */
int spyhip_wilson_cond(spyhip_ctx* ctx, const void* csd_c64_d, int nf, int n, double eps, void* A_d, void* work_d,
                       double* cond_out);
int spyhip_wilson_init(spyhip_ctx* ctx, const void* A_d, int nf, int n, int f_lo, int nftot, void* U_d,
                       void* gamma_part_d);
int spyhip_wilson_psi0(spyhip_ctx* ctx, void* gamma0_d, int n, int nf, void* psi0_d, void* psi_d);
int spyhip_wilson_g(spyhip_ctx* ctx, const void* psi_d, const void* U_d, int nf, int n, int pivoted, void* work_d,
                    void* g_d);
int spyhip_wilson_plus(spyhip_ctx* ctx, const void* g_d, int nftot, int64_t nent, void* gp_d, void* g0_d);
int spyhip_wilson_update(spyhip_ctx* ctx, void* psi_d, const void* gp_d, const void* g0_d, void* psi0_d,
                         const void* A_d, int nf, int n, void* work_d, double* err_out);
int spyhip_wilson_finish(spyhip_ctx* ctx, const void* A_d, const void* psi_d, const void* psi0_d, int nf, int n,
                         void* work_d, void* granger_d, void* H_d, void* Sigma_d);

/* Wilson iterations the last spyhip_granger call on this context ran (the reference's loop counter,
 * wilson_sf.py:77-109); for benchmarks and diagnostics. */
int spyhip_granger_last_iterations(const spyhip_ctx* ctx);

/* ---- utilities on the in-HBM trial queue ---------------------------------- */
/* out[r, :] = alpha * sum_t in[t, r, :]  (trial mean of (T, n) float32) */
int spyhip_trial_mean_f32(spyhip_ctx* ctx, const float* in_d, float* out_d, int64_t ntrials, int64_t n);
/* the same for (T, n) complex64: sums per component, then the reference's complex division by the real count - a
 * multiplication by the float32 reciprocal 1/T (NumPy's Smith division, summary_stats.py:426) */
int spyhip_trial_mean_c64(spyhip_ctx* ctx, const void* in_d, void* out_d, int64_t ntrials, int64_t n);
/* spy.mean(data, dim=<axis label>) for one trial (statistics/summary_stats.py:24, statistics/compRoutines.py:22-57:
 * np.nanmean(trial, axis, keepdims=True)): x_d (outer, n, inner) float32 or complex64 -> out_d (outer, inner); NaN
 * elements are skipped; float32 sums in NumPy's order (rows in order for an axis that is not the last, pairwise
 * blocks for the last one), one division in float64. */
int spyhip_axis_nanmean(spyhip_ctx* ctx, const void* x_d, int64_t outer, int64_t n, int64_t inner, int is_complex,
                        void* out_d);

#ifdef __cplusplus
}
#endif
#endif /* SPYHIP_H */
