/*
 * mi355zk.h -- C-ABI of libmi355zk.so: the MI355X (gfx950) BN254 G1-MSM / Fr-NTT hot path that slots in
 * under scroll-prover's `halo2_proofs` dependency.
 *
 * Boundary.  scroll-prover itself holds no proving arithmetic; its GPU builds replace the whole
 * `halo2_proofs` crate through a cargo path override [REF docker/chain-prover/gpu/Dockerfile:7-8],
 * [REF docker/trace-prover/gpu/Dockerfile:6-7], next to the existing `[patch]` redirect [REF Cargo.toml:40-41].
 * The functions below are what such a replacement crate's FFI binds (INTEGRATION.md shows the Rust `extern "C"`
 * block and the patched call sites).  Each entry point names the halo2_proofs / halo2curves function it
 * stands in for; those crates are pinned at scroll-tech/halo2@e5ddf67 and scroll-tech/halo2curves@112f5b9
 * [REF Cargo.lock:1886-1888,1911-1913] and reached from [REF integration/src/prove.rs:37,67,96].
 *
 * Data conventions (pinned by fixture KATs A1-A4):
 *   Fr, Fq       32 B   4 x u64 little-endian limbs, Montgomery form (R = 2^256), fully reduced
 *   G1Affine     64 B   {x, y};  identity = (0, 0)                       == halo2curves::bn256::G1Affine
 *   G1           96 B   Jacobian {x, y, z}; identity z = 0               == halo2curves::bn256::G1
 * Results returned as G1 are normalised: (x, y, R) with R = Montgomery one, or (0, 0, 0) for the identity,
 * so `to_affine()` on the Rust side is a no-op and byte comparison is meaningful.
 *
 * Pointer flavours: `*_host` arguments are ordinary process memory owned by the caller (Rust Vec<..>): the
 * library copies to HBM, computes and copies back before returning.  `*_dev` arguments are HIP device
 * pointers on a bound device: blocks from mi355_buf_alloc (what the Rust shim's DevicePoly holds) or any other device
 * allocation (torch tensors in the bench) -- for callers that keep polynomials resident, SURVEY §8f-1.
 *
 * Errors: every function returns 0 on success or one of MI355_E*; nothing throws, aborts or unwinds across
 * the boundary.  mi355_last_error() gives a thread-local message.  There is NO CPU fallback inside the
 * library: without a usable gfx950 device every compute entry point fails with MI355_ENODEVICE and the
 * caller (the Rust shim) decides what to do.
 *
 * Threading: entry points may be called from any thread (rayon workers); the bound device is re-selected on the calling thread.  Locks are PER
 * DEVICE: a call that works on one device (every transform, scan, evaluation, element-wise operation, buffer copy) holds that device's lock
 * only, so callers on different devices run concurrently; MSM entry points hold every bound device (the point range is sharded over all of
 * them); lifecycle and SRS management hold everything.  The MSM options (mi355_msm_set_normalise / _set_window_bits) are per calling
 * THREAD: a setting made on one thread does not affect MSMs issued from another.  mi355_profile_enable may be called before mi355_init.
 *
 * Devices: mi355_init(id) binds one device (one process per GPU).  mi355_init_multi(ids, n) binds n devices to ONE process -- the shape
 * of the reference, where one prover process holds one params_map [REF integration/src/prove.rs:11-21]: registered bases are then
 * sharded by point range over the devices, every MSM entry point fans out behind the same signature, the per-device partial sums are
 * exchanged with one ncclAllGather (RCCL over xGMI) and folded on the first device.  The NTT does not shard at k <= 26
 * ("replicas only"): a `*_dev` call runs on the device that owns its operands (mi355_buf_alloc(.., slot)), a `*_host` call on whichever
 * bound device is free (round-robin), and the `*_batch_*` transforms deal independent polynomials over all bound devices.
 */
#ifndef MI355ZK_H
#define MI355ZK_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MI355_OK 0
#define MI355_EBADARG 1
#define MI355_ENODEVICE 2
#define MI355_EOOM 3
#define MI355_EHIP 4
#define MI355_ERCCL 5

/* ---- lifecycle ------------------------------------------------------------------------------------------- */
/* Bind this process to HIP device `device_id` (ordinal within HIP_VISIBLE_DEVICES) and create the stream.    */
int mi355_init(int device_id);
/* SURVEY 8b's `mi355_init(const int *device_ids, int n_devices)`: bind n_devices (1..16) devices to this process, create one RCCL
 * communicator over them (ncclCommInitAll; librccl.so.1 is loaded on demand, MI355_ERCCL if that fails) and enable peer access.
 * device_ids[0] is the primary device.  Listing one physical device twice is a TEST mode (needs MI355_ALLOW_DUP_DEVICES=1: a one-GPU box
 * can then run the N = 2 control flow with real kernels; the exchange becomes a device copy because no communicator can span one device
 * twice).  MI355_MULTI_FORCE=1 sends even a one-device MSM through the partial + exchange + fold path (a real one-rank ncclAllGather).   */
int mi355_init_multi(const int *device_ids, int n_devices);
int mi355_device_count(int *n_out);
int mi355_shutdown(void);
const char *mi355_last_error(void);
const char *mi355_version(void);
/* Launch all kernels on `hip_stream` (a hipStream_t, e.g. torch's current stream).  NULL selects the HIP null
 * (legacy default) stream -- torch's default stream.  After mi355_init the library uses a private non-blocking
 * stream; mi355_reset_stream() returns to it.                                                                 */
int mi355_set_stream(void *hip_stream);
int mi355_reset_stream(void);
/* Block until everything queued by the library has finished.                                                  */
int mi355_synchronize(void);

/* ---- resident buffers: where a proof's polynomials live between the calls of create_proof (the reference's caller is one long
 *      create_proof per layer [REF integration/src/prove.rs:36-43,67,96]).  A Rust `DevicePoly` (rust_shim/mi355zk.rs) owns one block and frees
 *      it on Drop; every `*_dev` entry point accepts pointers into these blocks (at any offset) -- and any other HIP device pointer of a bound
 *      device -- and runs on the device that owns them.  device_slot indexes the list given to mi355_init_multi (0 = the primary / only device).
 *      mi355_buf_free returns the block to a pool (no hipFree, which would synchronise the device); work already queued on it stays valid.   */
int mi355_buf_alloc(uint64_t bytes, int device_slot, void **dev_ptr_out);
int mi355_buf_free(void *dev_ptr);
int mi355_buf_trim(void);                                   /* give every pooled (free) block back to HIP                                  */
int mi355_buf_slot(const void *dev_ptr, int *slot_out);     /* which device slot owns this pointer                                         */
/* host -> device on the owner's COPY stream: overlaps the compute queued before AND after it; an upload into a block no call has used since
 * mi355_buf_alloc waits only for the work queued on it before its last mi355_buf_free.  Synchronous: on return the data is in HBM (calls
 * issued afterwards see it) and the host buffer may be reused.  Takes no device lock.                                                   */
int mi355_buf_upload(void *dst_dev, const void *src_host, uint64_t bytes);
int mi355_buf_download(void *dst_host, const void *src_dev, uint64_t bytes);   /* ordered after everything queued on the owner; synchronous */
/* Narrow uploads for narrow columns (create_proof steps 2-3 are PCIe-bound at the many-column layers while most witness cells are zeros, bytes or 64-bit words: SURVEY 8d's
 * witness-like distribution; lookup_bits of [REF integration/configs/layer1.config:11]).  dst receives n 32-byte Montgomery words either way.
 *   packed  src = n little-endian unsigned integers of width_bytes in {1, 2, 4, 8} (CANONICAL values: selectors, byte / range-checked / limb columns, whose kind the caller
 *           knows statically); W bytes per cell cross the link, the device multiplies by R
 *   sparse  the non-zero cells as (index, 32-byte Montgomery value) pairs in any order; dst is zero-filled first; a pair whose index is >= n is dropped on the device (it never reaches memory).  mi355_host_compact_nonzero builds the pairs from a plain
 *           column with `threads` host threads (zero is zero in Montgomery form: the scan needs no arithmetic), idx_out / vals_out sized for n entries.
 * The narrow data crosses PCIe like mi355_buf_upload (copy stream, no device lock; the host buffers may be reused on return); the expansion is QUEUED on the owner's compute
 * stream: calls issued afterwards on that device see the data.                                                                                                          */
int mi355_buf_upload_packed(void *dst_dev, const void *src_host, uint64_t n, uint32_t width_bytes);
int mi355_buf_upload_sparse(void *dst_dev, uint64_t n, const uint32_t *idx_host, const void *vals_host, uint64_t count);
int mi355_host_compact_nonzero(const void *src_host, uint64_t n, uint32_t *idx_out, void *vals_out, uint64_t *count_out, int threads);
int mi355_buf_copy(void *dst_dev, const void *src_dev, uint64_t bytes);        /* within a device or between two bound devices (xGMI)       */
int mi355_buf_zero(void *dst_dev, uint64_t bytes);
/* Page-locked host memory for buffers that cross PCIe more than once, or once but on the critical path (the witness columns of the many-column layers):
 * a first copy out of ordinary (pageable) memory runs at ~34 GB/s on this platform, out of page-locked memory at the link's ~56 GB/s.  The caller writes
 * its column into the block (witness synthesis can write there directly) and passes it to mi355_buf_upload / any `*_host` entry point like any pointer. */
int mi355_host_alloc(uint64_t bytes, void **host_ptr_out);
int mi355_host_free(void *host_ptr);
/* HBM accounting of one bound device (any pointer may be NULL): what HIP reports free / in total, and what this library holds in live
 * mi355_buf blocks, in pooled (freed, reusable) blocks and in its grow-only workspace arena.  A prover keeps the SRS of its degree set
 * [REF bin/src/trace_prover.rs:35-36] AND the proving key's extended-coset polynomials resident: the caller budgets window tables
 * (mi355_srs_precompute: W x the basis) against this figure and stays on the table-free schedule when they do not fit.     */
int mi355_mem_info(int device_slot, uint64_t *free_bytes, uint64_t *total_bytes, uint64_t *live_buf_bytes, uint64_t *pooled_bytes, uint64_t *workspace_bytes);

/* ---- SRS ownership: ParamsKZG { g, g_lagrange } [halo2_proofs poly/kzg/commitment.rs], held for the process
 *      lifetime by the caller's params_map [REF bin/src/trace_prover.rs:35-43], [REF integration/src/prove.rs:12,26,58].
 *      A handle is one basis (n affine points) resident in HBM.                                               */
int mi355_srs_register_host(const void *bases_affine_host, uint64_t n, uint64_t *handle_out);
int mi355_srs_register_dev(const void *bases_affine_dev, uint64_t n, int copy, uint64_t *handle_out);
/* Prover::load_params for one degree [REF bin/src/trace_prover.rs:35-36; prover/src/utils.rs EXT-recalled]: a RawBytes params{k} file
 * (u32 LE k | g[2^k] x 64 B | g_lagrange[2^k] x 64 B | g2 128 B | s_g2 128 B; any other length is rejected, as load_params does)
 * streamed straight into device memory through two pinned staging buffers and registered as two library-owned bases.
 * flags bit 0: validate every point on the device (identity, or reduced coordinates on y^2 = x^3 + 3 -- the check SerdeFormat::RawBytes
 * makes on the CPU and RawBytesUnchecked skips).  g2_out / s_g2_out (optional, 128 B each) receive the two G2 points untouched.     */
int mi355_srs_load_params_file(const char *path, uint32_t flags, uint32_t *k_out, uint64_t *g_handle_out, uint64_t *g_lagrange_handle_out, void *g2_out, void *s_g2_out);
/* `&params.g[..n]` as a handle of its own (ParamsKZG::downsize keeps g[..2^k]; load_params_map clones + downsizes
 * [REF integration/tests/integration.rs:17-22]): the first n points of `parent_handle`, SHARING its device memory and window tables --
 * no copy, no second 48 GiB table.  Handles that share memory may be released in any order; the last one frees it.                 */
int mi355_srs_register_prefix(uint64_t parent_handle, uint64_t n, uint64_t *handle_out);
int mi355_srs_release(uint64_t handle);
/* Optional, once per basis: build T[w][i] = 2^(c w) * P_i (w < W = ceil(255 / c), affine, W * n * 64 B of HBM) so that all
 * windows of an MSM on this basis share ONE bucket set: no per-window Horner (255 serial doublings), W x fewer bucket
 * reductions.  c = 0 picks the window for MSMs of n_hint points (0 = the whole basis).  MSMs on the handle then use the
 * table whenever it is the cheaper schedule for their length.  The SRS is fixed for the life of the prover (params_map
 * [REF bin/src/trace_prover.rs:35-43]), so this is registration-time work, like the reference's own g_lagrange set-up. */
int mi355_srs_precompute(uint64_t handle, uint64_t n_hint, int c);
int mi355_srs_pre_dev_ptr(uint64_t handle, void **dev_ptr_out, int *c_out, int *windows_out);
int mi355_srs_len(uint64_t handle, uint64_t *n_out);
/* device pointer of the resident basis (n x 64 B), for tests and chained device-side work                     */
int mi355_srs_dev_ptr(uint64_t handle, void **dev_ptr_out);

/* ---- MSM: halo2_proofs::arithmetic::best_multiexp(coeffs, bases) -> C::Curve, and the two wrappers
 *      ParamsKZG::commit / commit_lagrange = best_multiexp(poly, &g[..n] / &g_lagrange[..n]).
 *      out_g1 receives 96 B (normalised Jacobian, see above) in host memory.                                  */
int mi355_msm_g1_host(uint64_t srs_handle, uint64_t base_offset, const void *scalars_host, uint64_t n, void *out_g1_host);
int mi355_msm_g1_dev(uint64_t srs_handle, uint64_t base_offset, const void *scalars_dev, uint64_t n, void *out_g1_host);
/* multi-GPU leg: the per-GPU partial sum stays in device memory (96 B at out_g1_dev, written in stream order, no host
 * synchronisation) so that it can be handed to the RCCL all-gather directly; mi355_g1_sum_dev folds the gathered partials.        */
int mi355_msm_g1_dev_async(uint64_t srs_handle, uint64_t base_offset, const void *scalars_dev, uint64_t n, void *out_g1_dev);
int mi355_g1_sum_dev(const void *points_jac_dev, uint64_t n, void *out_g1_host);
/* `batch` commitments over the SAME basis slice in one pass (e.g. all advice columns of a phase: create_proof commits them one
 * after the other): scalars_dev is a host array of `batch` device pointers (n x 32 B each), out receives
 * batch x 96 B.  Results are identical to `batch` separate calls; the fixed per-call costs are paid once.                          */
int mi355_msm_g1_batch_dev(uint64_t srs_handle, uint64_t base_offset, const void *const *scalars_dev, uint32_t batch, uint64_t n, void *out_g1_host);
/* the same with `batch` host pointers (what the Rust loop over advice columns holds)                                              */
int mi355_msm_g1_batch_host(uint64_t srs_handle, uint64_t base_offset, const void *const *scalars_host, uint32_t batch, uint64_t n, void *out_g1_host);
/* ad-hoc bases (best_multiexp with bases that are not a registered SRS)                                        */
int mi355_msm_g1_adhoc_host(const void *bases_affine_host, const void *scalars_host, uint64_t n, void *out_g1_host);
/* sum of `n` G1 (Jacobian, any representative) points: the fold `results.iter().fold(identity, |a, b| a + b)` of
 * best_multiexp, used to combine per-GPU partial sums after the RCCL all-gather.                  */
int mi355_g1_sum_host(const void *g1_points_host, uint64_t n, void *out_g1_host);
/* group::Curve::batch_normalize(p: &[G1], q: &mut [G1Affine]) [EXT-recalled halo2curves bn256 / group crate; create_proof normalises each
 * vector of projective commitments with it before they enter the transcript]: n Jacobian points (96 B, any representative) -> n affine
 * points (64 B), the identity (Z = 0) becomes (0, 0).  One shared inversion per 256 points (Montgomery's trick).  Input and output
 * must not overlap; the _dev variant is asynchronous on the library stream.                                                          */
int mi355_g1_batch_normalize_dev(const void *g1_points_dev, void *affine_out_dev, uint64_t n);
int mi355_g1_batch_normalize_host(const void *g1_points_host, void *affine_out_host, uint64_t n);
/* (per calling thread) normalise = 0: subsequent MSM results are SOME Jacobian representative of the sum (as best_multiexp's C::Curve is) instead of the
 * normalised one; saves the serial field inversion (~0.4 ms) where the result is folded again anyway (per-GPU partial sums).      */
int mi355_msm_set_normalise(int on);
/* tuning (per calling thread): window bits c for subsequent MSMs: 0 = automatic from n (window tables used when registered and cheaper),
 * -1 = automatic but ignoring window tables (the memory-lean schedule: per-window bucket sets + Horner), 2..24 = fixed, no tables       */
int mi355_msm_set_window_bits(int c);
/* pipelined schedule of a large single MSM: the point range is cut into `chunks` slices and the (memory-bound) sort of slice k + 1
 * runs under the (ALU-bound) accumulation of slice k on separate HIP streams; results are identical.  Off by default (measured
 * slower on MI355X: the accumulation holds every wave slot; env MI355_MSM_CHUNKS).  chunks = 1 disables, 0 restores the default;
 * min_log_n = smallest log2(n) that is cut.                                                                                       */
int mi355_msm_set_pipeline(uint32_t chunks, uint32_t min_log_n);

/* ---- NTT: halo2_proofs::arithmetic::best_fft(a, omega, log_n): in place, natural order in -> natural order out,
 *      a'[i] = sum_j a[j] omega^(ij), no scaling.  omega: 32 B Montgomery, must have order 2^log_n.            */
int mi355_ntt_fr_host(void *data_host, uint32_t log_n, const void *omega);
int mi355_ntt_fr_dev(void *data_dev, uint32_t log_n, const void *omega);
/* EvaluationDomain::ifft = best_fft(a, omega_inv, log_n) followed by a[i] *= divisor (= n^-1) [poly/domain.rs]   */
int mi355_intt_fr_host(void *data_host, uint32_t log_n, const void *omega_inv, const void *divisor);
int mi355_intt_fr_dev(void *data_dev, uint32_t log_n, const void *omega_inv, const void *divisor);
/* EvaluationDomain::coeff_to_extended: dst[2^log_ext] = fft_{extended_omega}( zero-pad(coeffs[2^log_n]) with
 * a[i] *= {1, g_coset, g_coset_inv}[i % 3] )  (distribute_powers_zeta, into the coset)                          */
int mi355_coeff_to_extended_host(void *dst_host, const void *coeffs_host, uint32_t log_n, uint32_t log_ext,
                                 const void *g_coset, const void *g_coset_inv, const void *extended_omega);
int mi355_coeff_to_extended_dev(void *dst_dev, const void *coeffs_dev, uint32_t log_n, uint32_t log_ext,
                                const void *g_coset, const void *g_coset_inv, const void *extended_omega);
/* EvaluationDomain::extended_to_coeff: in place ifft(extended_omega_inv, extended_ifft_divisor) then
 * a[i] *= {1, g_coset_inv, g_coset}[i % 3]; the caller truncates.                                               */
int mi355_extended_to_coeff_host(void *data_host, uint32_t log_ext, const void *g_coset, const void *g_coset_inv,
                                 const void *extended_omega_inv, const void *extended_ifft_divisor);
int mi355_extended_to_coeff_dev(void *data_dev, uint32_t log_ext, const void *g_coset, const void *g_coset_inv,
                                const void *extended_omega_inv, const void *extended_ifft_divisor);

/* distribute_powers: a[i] *= factor^i (in place), and the general coset transform dst = best_fft(coeffs[i] * coset_factor^i, omega):
 * what `coeff_to_extended_part(poly, g_coset * extended_omega^j)` of the scroll fork does per quotient part [EXT-recalled domain.rs]. */
int mi355_distribute_powers_fr_dev(void *data_dev, uint64_t n, const void *factor);
int mi355_coset_ntt_fr_dev(void *dst_dev, const void *coeffs_dev, uint32_t log_n, const void *coset_factor, const void *omega);
/* `batch` independent transforms in one call -- the loops of create_proof over columns (the iNTT of every advice column, the
 * coset NTTs of every polynomial that enters evaluate_h; 8 + 32 for a layer-4 proof [REF integration/configs/layer4.config:3-10]).
 * divisor == NULL: best_fft; divisor = n^-1: EvaluationDomain::ifft.  Host pointers are dealt round-robin over the bound devices (one
 * worker thread and PCIe link per device; on each device the upload of item i + 1, the transform of item i and the download of item
 * i - 1 overlap over two staging buffers: 56 ms instead of 87 ms per 2^26 polynomial); device pointers run on the device that owns
 * them, concurrently across devices.  Results equal the serial loop's.                                                              */
int mi355_ntt_fr_batch_host(void *const *data_host, uint32_t batch, uint32_t log_n, const void *omega, const void *divisor);
int mi355_ntt_fr_batch_dev(void *const *data_dev, uint32_t batch, uint32_t log_n, const void *omega, const void *divisor);
int mi355_coset_ntt_fr_batch_dev(void *const *dst_dev, const void *const *coeffs_dev, uint32_t batch, uint32_t log_n, const void *coset_factor, const void *omega);

/* element-wise operations on device-resident vectors of Fr (op 0: a + b, 1: a - b, 2: a * b; dst may alias a or b) and
 * data[i] *= table[i mod period] (period a power of two <= 4096; EvaluationDomain::divide_by_vanishing_poly multiplies the extended
 * evaluations by the inverted t_evaluations, whose period is 2^(extended_k - k)).  The pointwise glue of SURVEY 8f-1.            */
int mi355_fr_vec_op_dev(int op, void *dst_dev, const void *a_dev, const void *b_dev, uint64_t n);
/* dst = a + scalar * b (a_dev == NULL: dst = scalar * b; dst may alias a or b): the running linear combination sum_i v^i p_i(X) of the
 * multi-open argument and every other "poly * scalar" of create_proof [EXT-recalled halo2_proofs poly: Polynomial * F, + ].       */
int mi355_fr_vec_axpy_dev(void *dst_dev, const void *a_dev, const void *b_dev, const void *scalar, uint64_t n);
int mi355_fr_vec_mul_periodic_dev(void *data_dev, uint64_t n, const void *table_host, uint32_t period);
/* The operand shape of evaluate_h [EXT-recalled halo2_proofs src/plonk/evaluation.rs; SURVEY 3.2 step 7: gates read ROTATED columns,
 * a[(i + rot * 2^(extended_k - k)) mod n], and evaluate an expression over dozens of extended-domain polynomials]:
 *     dst[i] (+)= sum_{j < n_terms} coeffs[j] * prod_{k < term_len[j]} polys[factor_poly[.]][(i + factor_rot[.]) mod n]
 * in ONE launch instead of a chain of mi355_fr_vec_op_dev calls (each a full HBM round trip).  Term j owns term_len[j] consecutive entries
 * of factor_poly / factor_rot (host arrays); coeffs: n_terms x 32 B Montgomery (host); rotations in ELEMENTS, already scaled by the caller,
 * negative values allowed; n a power of two (the extended domain, or one 2^k coset part).  Limits per launch: 24 polynomials, 16 terms,
 * 16 factors per term, 48 factors in all (larger expressions are split, accumulate = 1 adds to dst).  dst may alias a polynomial only
 * when every rotation of that polynomial is zero.  Terms with c_j = 1 or c_j = -1 cost no multiplication for the coefficient (2^26, 9 terms /
 * 20 factors: 8.7 ms against 12.2 ms with general coefficients).                                                                                   */
int mi355_fr_gate_eval_dev(void *dst_dev, const void *const *polys_dev, uint32_t n_polys, const void *coeffs_fr_host, const uint32_t *term_len,
                           uint32_t n_terms, const uint32_t *factor_poly, const int32_t *factor_rot, uint64_t n, int accumulate);
/* the extended-domain vector from its Q <= 8 coset parts (scroll fork: evaluate_h works part by part, part q = the evaluations at
 * zeta * extended_omega^(q + Q i), i < n): dst[i * Q + q] = parts[q][i], i.e. the natural order mi355_extended_to_coeff_dev inverts           */
int mi355_fr_interleave_dev(void *dst_dev, const void *const *parts_dev, uint32_t q_parts, uint64_t n);
/* the multiplicative scans of the permutation / lookup arguments [EXT-recalled halo2_proofs src/plonk/permutation/prover.rs,
 * src/plonk/lookup/prover.rs]: data[i] = data[i]^-1 with zeros left zero (ff::BatchInvert), and the grand product
 * dst[0] = 1, dst[i] = prod_{j<i} src[j] (dst may alias src; total_out_host, optional, receives prod_{j<n} src[j] and makes the
 * call synchronous).  Asynchronous on the library stream otherwise.                                                               */
int mi355_fr_batch_invert_dev(void *data_dev, uint64_t n);
/* kate_division(poly, z) = (poly(X) - poly(z)) / (X - z) [EXT-recalled halo2_proofs src/arithmetic.rs], the quotient polynomials of the
 * multi-open argument: n coefficients in, n - 1 out (q_i = a_(i+1) + z q_(i+1), a parallel scan).  dst must not overlap poly, except
 * dst == poly + 1 element (the quotient then replaces coefficients 1 .. n-1 in place).                                             */
int mi355_fr_kate_division_dev(void *dst_dev, const void *poly_dev, uint64_t n, const void *z);
int mi355_fr_prefix_product_dev(void *dst_dev, const void *src_dev, uint64_t n, void *total_out_host);
/* the additive counterpart: dst[0] = 0, dst[i] = sum_{j<i} src[j] -- the running sum phi of the log-derivative (mv-lookup) argument of the
 * scroll fork [EXT-recalled halo2_proofs src/plonk/mv_lookup/prover.rs: phi[i + 1] = phi[i] + sum_j 1 / (beta + f_j[i]) - m[i] / (beta + t[i]);
 * SURVEY 3.2 step 4 "lookup grand-sum"].  Same aliasing and total_out_host rules (the total must be zero for a valid argument).       */
int mi355_fr_prefix_sum_dev(void *dst_dev, const void *src_dev, uint64_t n, void *total_out_host);

/* ---- halo2_proofs::arithmetic::eval_polynomial(poly, point) = sum_i poly[i] * point^i  (the evaluations written to the
 *      transcript in step 9 of create_proof); out_fr_host receives 32 B.  First widening into SURVEY 8f-3.   */
int mi355_eval_polynomial_dev(const void *poly_dev, uint64_t n, const void *point, void *out_fr_host);
/* `batch` evaluations with one device synchronisation (step 9 of create_proof: every queried (polynomial, rotation) pair): polys_dev[i] has n
 * coefficients, points = batch x 32 B (Montgomery), out_fr_host = batch x 32 B.  Same values as `batch` calls of mi355_eval_polynomial_dev.      */
int mi355_eval_polynomial_batch_dev(const void *const *polys_dev, uint32_t batch, uint64_t n, const void *points, void *out_fr_host);
int mi355_eval_polynomial_host(const void *poly_host, uint64_t n, const void *point, void *out_fr_host);

/* ---- synthetic SRS: ParamsKZG::setup(k, rng) restated on the device [poly/kzg/commitment.rs]:
 *      g[i] = tau^i G,  g_lagrange[i] = L_i(tau) G.  Writes n = 2^k affine points per basis to device memory.   */
int mi355_srs_setup_dev(void *g_dev, void *g_lagrange_dev, uint32_t k, const void *tau, const void *omega);
/* points[i] = scalars[i] * G (fixed-base, batch-normalised); building block of the above                      */
int mi355_g1_fixed_base_mul_dev(void *points_affine_dev, const void *scalars_dev, uint64_t n);

/* ---- G2: out = scalar * p on the twist y^2 = x^3 + 3 / (9 + u) over Fq2.  Points are 128-byte halo2curves G2Affine values
 * (x.c0 | x.c1 | y.c0 | y.c1, Montgomery limbs; identity = all zero) -- the layout of `g2` / `s_g2` in a RawBytes params file.  The one G2
 * operation on the path: ParamsKZG::setup's s_g2 = tau * G2 [EXT-recalled poly/kzg/commitment.rs]; a G2 MSM does not exist in create_proof
 * (SURVEY 8a a7).  MI355_EBADARG when p is not on the twist.  The generator constant is halo2curves' G2 generator, pinned by the
 * pairing input of the released verifier [REF release-v0.13.1/evm_verifier.yul:1230-1233].                                             */
int mi355_g2_mul_host(const void *p_g2affine_host, const void *scalar_fr, void *out_g2affine_host);

/* ---- best_fft::<Fr, G1> -- the same DFT over G1 points, a'[i] = sum_j omega^(ij) a[j] -- and its one caller, g_to_lagrange, which
 * ParamsKZG::downsize(k) [REF integration/tests/integration.rs:17-22] and ParamsKZG::setup run to rebuild g_lagrange from
 * g[..2^k]: g_lagrange = n^-1 * DFT_{omega^-1}(g) [EXT-recalled halo2_proofs src/arithmetic.rs g_to_lagrange].
 * Points: 96-byte Jacobian in place (any representative in, normalised z = R / all-zero identity out); g_to_lagrange takes and
 * returns 64-byte affine points (input and output may be the same buffer).                                                        */
int mi355_g1_fft_dev(void *points_jac_dev, uint32_t log_n, const void *omega);
int mi355_g1_fft_host(void *points_jac_host, uint32_t log_n, const void *omega);
int mi355_g_to_lagrange_dev(const void *g_affine_dev, void *g_lagrange_affine_dev, uint32_t log_n, const void *omega_inv, const void *n_inv);
/* downsize on handles: a NEW library-owned basis g_lagrange' = g_to_lagrange(g[..2^k]) from a registered coefficient basis
 * (omega_inv = omega_k^-1, n_inv = 2^-k, Montgomery); mi355_srs_read_host copies points of any registered basis back, which is
 * how the Rust ParamsKZG refills its g_lagrange Vec after downsize.                                                                */
int mi355_srs_downsize(uint64_t g_handle, uint32_t k, const void *omega_inv, const void *n_inv, uint64_t *g_lagrange_handle_out);
int mi355_srs_read_host(uint64_t handle, uint64_t offset, uint64_t n, void *out_affine_host);

/* ---- measurement hooks (bench.py): HIP-event timing of the kernels of the most recent MSM / NTT call.       */
int mi355_profile_enable(int on);
/* name in {"msm_total","msm_digits","msm_sort","msm_accumulate","msm_reduce","ntt_total","ntt_pass"}; returns the
 * accumulated milliseconds and launch count since the last mi355_profile_reset().                             */
int mi355_profile_get(const char *name, double *ms_out, uint64_t *launches_out);
int mi355_profile_reset(void);
/* (c, windows, entries) chosen by the last MSM (on the primary device), for G1-adds accounting                 */
int mi355_msm_last_plan(int *c_out, int *windows_out, uint64_t *entries_out);
/* how the last MSM ran: device slots that took part; the exchange: "none" | "rccl_allgather" | "device_copy" (static string); whether the
 * window tables (one shared bucket set) were used on the primary device; point-range slices of the host-pointer path (1 = one copy)  */
int mi355_msm_last_run(int *devices_out, const char **exchange_out, int *shared_tables_out, int *host_slices_out);

/* test hook: copy `bytes` of the internal workspace buffer `role` (e.g. "msm.offsets", "msm.sorted") to the host   */
int mi355_debug_ws_read(const char *role, uint64_t offset, void *dst_host, uint64_t bytes);

#ifdef __cplusplus
}
#endif
#endif
