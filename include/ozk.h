/* ozk.h — C ABI of the MI355X-native BN254 MSM / FFT back end.
 *
 * This is the drop-in boundary: one shared library,
 * libozk_hip.so, with plain-C entry points (pointers + sizes, no torch / HIP
 * types).  Each `*_host` function takes exactly the byte buffers the reference's
 * JNI native receives and returns exactly the bytes it returns; the three JNI
 * shim libraries (include/ozk_jni.h) only move bytes between the JVM and these.
 * Each `*_dev` function is the same operation on buffers already resident in HBM
 * (device pointers), asynchronous on `stream`, for callers that keep data on the
 * GPU (bench.py, the prove harness, torch.distributed ranks).
 *
 * Wire formats (all integers canonical, NON-Montgomery, value < modulus):
 *   scalar / Fr element in : 32 B little-endian  (VariableBaseMSM.java:121-131)
 *   G1 point in            : X|Y|Z, 3 x 32 B LE  (VariableBaseMSM.java:221-228)
 *   G2 point in            : X.c0|X.c1|Y.c0|Y.c1|Z.c0|Z.c1, 6 x 32 B LE
 *                                                (bn254a/BN254aG2.java:77-86)
 *   var-MSM / FFT out      : 64 B LE per coordinate, upper 32 B zero
 *                                                (VariableBaseMSM.java:239-258)
 *   fixed-base / field out : 64 B BIG-endian per coordinate
 *                                                (algebra_msm_FixedBaseMSM.cu:783-787)
 * Returned points are affine-normalised Jacobian triples (X/Z^2, Y/Z^3, 1); the
 * point at infinity is (0, 1, 0) as BNG1.toAffineCoordinates (BNG1.java:163-172).
 *
 * Every function returns 0 on success and a negative OZK_E_* code on failure;
 * ozk_last_error() gives the message for the calling thread.  Nothing here ever
 * falls back to a CPU path: without a usable GPU the calls fail with
 * OZK_E_NO_DEVICE.
 */
#ifndef OZK_H
#define OZK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OZK_OK 0
#define OZK_E_INVALID -1    /* bad argument (null pointer, n <= 0, not a power of two ...) */
#define OZK_E_NO_DEVICE -2  /* no HIP device / device call failed */
#define OZK_E_NOMEM -3      /* device or host allocation failed */
#define OZK_E_INTERNAL -4

#define OZK_G1 1 /* `type` / `BNType` value for G1; anything else is G2, as in the reference */
#define OZK_G2 2

const char* ozk_last_error(void);
int ozk_version(void);
/* number of visible HIP devices (0 if none); `taskID % count` selects the device as
 * the reference does (algebra_msm_VariableBaseMSM.cu:1249-1257). */
int ozk_device_count(void);
/* Streams confined to compute units [first_cu, first_cu + n_cus) of the current device (hipExtStreamCreateWithCUMask;
 * ozk_device_cu_count() = the device's compute units, 256 on an MI355X), for callers that partition the chip between
 * the stages of consecutive MSMs (device.VarMsmPipeline3(tail_cus=...): tails on a few units, the accumulation on the
 * rest).  Such a stream is a BLOCKING stream: it synchronises with the null stream, so nothing of a schedule that
 * uses it may run there.  ozk_stream_destroy gives it back. */
int ozk_device_cu_count(void);
int ozk_stream_create_cu_range(int32_t first_cu, int32_t n_cus, void** stream);
int ozk_stream_destroy(void* stream);
/* The OZK_* tuning environment variables are read once per process and cached (a plan must not change
 * between the workspace-size query and the run); tests and tuning scripts call this after changing one. */
int ozk_tuning_reload(void);
/* The `*_host` entry points keep their streams, device arena and pinned staging buffers in a per-device
 * pool between calls (the reference allocates and frees inside every native call,
 * algebra_msm_VariableBaseMSM.cu:1292-1303,1405-1409).  This gives everything back. */
int ozk_host_cache_release(void);
/* Where the wall time of the calling thread's last `*_host` call went: stats10 = {context acquire, arena growth,
 * waits for a pinned staging buffer, host memcpy into the ring, host memcpy out of it, enqueueing copies,
 * stream synchronisation} in milliseconds, the number of staging waits and of host memcpys, and the call's total
 * time as the library saw it (entry to context release).  (How a call that takes several times its median is
 * attributed — to one of the library's waits, or to the caller's side of the boundary — tools/host_jitter.py.) */
int ozk_host_call_stats(double* stats10);

/* ---------------- VariableBaseMSM ---------------------------------------
 * replaces Java_algebra_msm_VariableBaseMSM_variableBaseSerialMSMNativeHelper
 * (algebra_msm_VariableBaseMSM.h:13-16, .cu:1614-1695).
 * bases: n x 96 B (G1) or n x 192 B (G2); scalars: n x 32 B; out: 192 B / 384 B. */
int ozk_var_msm_host(const uint8_t* bases, const uint8_t* scalars, int32_t n, int32_t type,
                     int32_t task_id, uint8_t* out);
/* replaces ..._variableBaseDoubleMSMNativeHelper (.h:21-24, .cu:1712-1788).
 * out: 576 B = G1 result (192) || G2 result (384). */
int ozk_var_double_msm_host(const uint8_t* bases_g1, const uint8_t* bases_g2,
                            const uint8_t* scalars, int32_t n, int32_t task_id, uint8_t* out);

/* The same MSM spread over several GPUs from ONE call (no counterpart in the reference, whose native uses the
 * one device `taskID % count` selects, algebra_msm_VariableBaseMSM.cu:1249-1257; its multi-GPU form is one
 * Spark partition per device, VariableBaseMSM.java:775-786): the index range is cut into `shards` contiguous
 * slices (<= 0: one per visible device), slice i runs on device i % count from its own host thread, the
 * partial results are added on device 0.  Same bytes out as ozk_var_msm_host.  ozk_var_msm_auto_host is what
 * the JNI shim calls: ozk_var_msm_host on device taskID % count, as the reference; with OZK_SHARD=1 (opt-in, for a
 * serial prover that is the only caller) calls of n >= OZK_SHARD_MIN_N (2^21) pairs are sharded over the visible
 * devices instead.  Unverified on multi-GPU hardware (every box seen so far had one GPU). */
int ozk_var_msm_sharded_host(const uint8_t* bases, const uint8_t* scalars, int32_t n, int32_t type, int32_t shards,
                             uint8_t* out);
int ozk_var_msm_auto_host(const uint8_t* bases, const uint8_t* scalars, int32_t n, int32_t type, int32_t task_id,
                          uint8_t* out);
/* The double MSM the same way (VariableBaseMSM.distributedDoubleMSM, VariableBaseMSM.java:805-818: per-partition
 * doubleMSM + reduce(add)): out = 576 B, G1 (192) || G2 (384); ozk_var_double_msm_auto_host is what the JNI native
 * calls (ozk_var_double_msm_host unless OZK_SHARD=1, as above). */
int ozk_var_double_msm_sharded_host(const uint8_t* bases_g1, const uint8_t* bases_g2, const uint8_t* scalars, int32_t n,
                                    int32_t shards, uint8_t* out);
int ozk_var_double_msm_auto_host(const uint8_t* bases_g1, const uint8_t* bases_g2, const uint8_t* scalars, int32_t n,
                                 int32_t task_id, uint8_t* out);
/* The partial results of a sharded call meet through an RCCL all-gather over xGMI (communicators made once per process
 * by ncclCommInitAll over the devices in use; librccl is bound at run time) followed by a HIP point sum on device 0 —
 * or, when RCCL is absent, fails to initialise, is busy with another sharded call or OZK_SHARD_RCCL=0, through the
 * host.  Same bytes either way.  ozk_shard_last_exchange(): how the calling thread's last sharded call did it
 * (1 = RCCL, 0 = host, -1 = a single shard / no call yet).  ozk_shard_comms_release() drops the communicators. */
int ozk_shard_last_exchange(void);
void ozk_shard_comms_release(void);

/* Device-resident variants.  `workspace` must hold ozk_var_msm_workspace_bytes(n, type)
 * bytes; all pointers are device pointers; `stream` is a hipStream_t (NULL = default). */
size_t ozk_var_msm_workspace_bytes(int32_t n, int32_t type);
int ozk_var_msm_dev(const void* d_bases, const void* d_scalars, int32_t n, int32_t type,
                    void* d_out, void* d_workspace, size_t workspace_bytes, void* stream);
/* The same MSM in two phases, for callers that keep several MSMs in flight (a Groth16 prove
 * has six independent ones): the HEAD is the throughput-bound part (sort, bucket accumulation,
 * first window-sum level) and leaves its result in `d_tail` (ozk_var_msm_tail_bytes bytes);
 * the TAIL is the latency-bound remainder (upper window-sum levels, Horner over the windows,
 * normalisation; a few lanes busy for ~2 ms) and reads nothing but `d_tail`.  Running the tail
 * on a second stream lets the next MSM's head (which may reuse the same workspace) overlap
 * it.  ozk_var_msm_dev == head + tail on one stream. */
/* ---- prepared bases (no counterpart in the reference, which re-marshals and
 * re-uploads the proving key for every MSM, VariableBaseMSM.java:224-227).  The bases are converted once
 * to the affine Montgomery records the accumulation kernel gathers (with the GLV plan: both halves,
 * 2n x 64 | 128 B) and stay in HBM; an MSM over them skips the conversion and the 96 | 192 B per base of
 * host-to-device traffic.  Results are byte-identical to ozk_var_msm_host on the same inputs.
 *   host form : handle = create(bases) ; msm(handle, scalars) any number of times ; destroy(handle).
 *               A handle serialises its MSMs (internal mutex); use one handle per concurrent caller.
 *   device form: ozk_var_msm_prepare_dev once, then ozk_var_msm_prepared_dev / _head_prepared_dev. */
int ozk_bases_create_host(const uint8_t* bases, int32_t n, int32_t type, int32_t task_id, void** handle);
int ozk_var_msm_bases_host(void* handle, const uint8_t* scalars, int32_t n, uint8_t* out);
int ozk_bases_destroy(void* handle);
/* OZK_G1 / OZK_G2 for a live handle, 0 for a stale or released one.  A handle is a token into a generation-
 * checked table, not a pointer: a late call with a released (or forged) handle fails with OZK_E_INVALID, and
 * ozk_bases_destroy really frees everything — at once, or when the MSM still running on the handle returns. */
int ozk_bases_type(void* handle);
size_t ozk_var_msm_prepared_bytes(int32_t n, int32_t type);
int ozk_var_msm_prepare_dev(const void* d_bases, int32_t n, int32_t type, void* d_prepared, size_t prepared_size,
                            void* stream);
int ozk_var_msm_prepared_dev(const void* d_prepared, const void* d_scalars, int32_t n, int32_t type, void* d_out,
                             void* d_workspace, size_t workspace_bytes, void* stream);
int ozk_var_msm_head_prepared_dev(const void* d_prepared, const void* d_scalars, int32_t n, int32_t type,
                                  void* d_workspace, size_t workspace_bytes, void* d_tail, size_t tail_bytes,
                                  void* stream, void* previous_levels_done);

/* Ordering hint for several MSMs in flight on two streams.  `tail_ordered` records `levels_done` after its first
 * window-sum level (after the last multi-wave level with OZK_MSM_ORDER_EARLY=0, round 1's form); `head_ordered`
 * waits for it after its sort and before its bucket accumulation, so that the previous MSM's wave-cooperative
 * level is resident before the accumulation takes three of the four wave slots of every SIMD.  Measured at 2^20:
 * 576 Mscalar-mul/s (538 with the late event; 576 with no event at all — the hint no longer buys throughput
 * since the accumulation kernel leaves a slot free, it only keeps the order deterministic).
 * Events come from ozk_order_event_create (a HIP event underneath). */
int ozk_order_event_create(void** ev);
int ozk_order_event_destroy(void* ev);
int ozk_var_msm_head_ordered_dev(const void* d_bases, const void* d_scalars, int32_t n, int32_t type,
                                 void* d_workspace, size_t workspace_bytes, void* d_tail, size_t tail_bytes,
                                 void* stream, void* previous_levels_done);
int ozk_var_msm_tail_ordered_dev(int32_t n, int32_t type, void* d_tail, size_t tail_bytes, void* d_out, void* stream,
                                 void* levels_done);
/* The tail's window sums come in two shapes (csrc/msm_var_driver.cuh, tail_shape): LATENCY (mode 0: fused first
 * level + wave-cooperative levels, the fewest dependent additions; what ozk_var_msm_dev, the host entry points and
 * the ordered tail above use) and THROUGHPUT (mode 1: serial levels, 3.0 instead of 5.25 additions per bucket,
 * ~40 dependent additions longer; what ozk_var_msm_tail_dev uses).  A caller that keeps the chip busy with other
 * work — a prover with five MSMs and a witness map in flight — picks throughput: every addition saved is vector-ALU
 * time for something else (a 2^20-constraint proof: 16.6 -> 16.3 ms).  levels_done may be NULL. */
int ozk_var_msm_tail_mode_dev(int32_t n, int32_t type, void* d_tail, size_t tail_bytes, void* d_out, void* stream,
                              void* levels_done, int32_t mode);

/* The head itself has two stages that stress different units — SORT (base conversion, digits,
 * counting sort: HBM / LDS) and ACCUMULATE (bucket accumulation ... first window-sum level:
 * vector ALU) — so a caller may pipeline three stages (sort of MSM k+2 | accumulate of k+1 |
 * tail of k).  The hand-off between them is the "sorted set" (double-buffer it); sort and
 * accumulate each have private scratch.  ozk_var_msm_head_dev == sort + accumulate.  (On
 * MI355X the three-stage form measures 562-574 Mscalar-mul/s at 2^20 against 576 for head | tail: the
 * sort kernels cannot co-reside with three accumulation blocks per CU and stretch the accumulation
 * when they can — profiles/r02_schedule_experiments.txt.) */
int ozk_var_msm_stage_bytes(int32_t n, int32_t type, size_t* sorted_bytes, size_t* sort_ws_bytes,
                            size_t* accum_ws_bytes);
int ozk_var_msm_sort_dev(const void* d_bases, const void* d_scalars, int32_t n, int32_t type,
                         void* d_sorted, size_t sorted_bytes, void* d_sort_ws, size_t sort_ws_bytes,
                         void* stream);
int ozk_var_msm_accum_dev(int32_t n, int32_t type, void* d_sorted, size_t sorted_bytes,
                          void* d_accum_ws, size_t accum_ws_bytes, void* d_tail, size_t tail_bytes,
                          void* stream);
/* the same two stages over prepared bases (ozk_var_msm_prepare_dev): the sort skips the base conversion */
int ozk_var_msm_sort_prepared_dev(const void* d_prepared, const void* d_scalars, int32_t n, int32_t type,
                                  void* d_sorted, size_t sorted_bytes, void* d_sort_ws, size_t sort_ws_bytes,
                                  void* stream);
int ozk_var_msm_accum_prepared_dev(const void* d_prepared, int32_t n, int32_t type, void* d_sorted,
                                   size_t sorted_bytes, void* d_accum_ws, size_t accum_ws_bytes, void* d_tail,
                                   size_t tail_bytes, void* stream);
/* ACCUMULATE in two parts, for a caller that keeps level 1 (the vector-ALU-bound kernel) back to back on one stream
 * and runs the rest (run merge, the short generic levels, the copy of the bucket counts: ~0.1 ms of low-occupancy work)
 * elsewhere: part 1 = level 1 only, part 2 = the rest (same arguments; it reads the sorted set and the accumulate
 * scratch level 1 wrote, so neither may be reused before it has run), part 0 = both (= the two entry points above).
 * d_prepared may be NULL (bases converted by the sort). */
int ozk_var_msm_accum_part_dev(const void* d_prepared, int32_t n, int32_t type, void* d_sorted, size_t sorted_bytes,
                               void* d_accum_ws, size_t accum_ws_bytes, void* d_tail, size_t tail_bytes,
                               void* stream, int32_t part);
size_t ozk_var_msm_head_workspace_bytes(int32_t n, int32_t type);
size_t ozk_var_msm_tail_bytes(int32_t n, int32_t type);
int ozk_var_msm_head_dev(const void* d_bases, const void* d_scalars, int32_t n, int32_t type,
                         void* d_workspace, size_t workspace_bytes, void* d_tail, size_t tail_bytes,
                         void* stream);
int ozk_var_msm_tail_dev(int32_t n, int32_t type, void* d_tail, size_t tail_bytes, void* d_out,
                         void* stream);

/* sum of k affine-normalised partial results in wire-out format (k x 192 B / 384 B), as
 * produced by ozk_var_msm_dev on k ranks -> one normalised point.  The multi-GPU
 * reduce(GroupT::add) of VariableBaseMSM.java:777-783 after the RCCL all-gather. */
int ozk_points_sum_dev(const void* d_points, int32_t k, int32_t type, void* d_out, void* stream);

/* Measurement hooks (bench.py): timing of the dominant kernel (the level-1 bucket accumulation,
 * k_segreduce<.., true>) per launch.  ozk_prof_enable(2): the kernel's own waves stamp the device's constant-rate
 * clock (first wave start -> last wave end), which leaves the schedule untouched; ozk_prof_enable(1): HIP start /
 * stop events on the dispatch (16 + k: on every (k+1)-th launch only) — exact too, but an event-carrying dispatch
 * costs the three-stage schedule 4-13 % of its throughput, so bench.py uses it as a cross-check in a second pass;
 * 0: off (the recorded launches stay readable).  ozk_prof_dominant_kernel_ms returns the mean duration over the
 * launches since the last enable, _stats the distribution.  ozk_var_msm_plan reports the window size the library
 * picks for n. */
int ozk_prof_enable(int on);
int ozk_prof_dominant_kernel_ms(double* avg_ms, int* launches);
/* stats4 = {mean, median, min, max} in ms (the box-to-box spread of the pool is ~10 %, so a single mean cannot
 * tell a 5 % gain from a slower box) */
int ozk_prof_dominant_kernel_stats(double* stats4, int* launches);
/* stats4 = {mean, median, min, max} over the launches recorded since ozk_prof_enable(2) of the SHADER clock in MHz
 * each launch ran at (shader-clock / constant-rate ticks stamped inside the kernel): what makes kernel times of two
 * boxes or two rounds comparable.  Profiling is single-device and serialised: launches on other devices than the one
 * current at ozk_prof_enable are not recorded; concurrent callers are safe. */
int ozk_prof_dominant_kernel_clock_mhz(double* stats4, int* launches);
/* ticks per millisecond of the device clock, calibrated against the host's steady clock by the first
 * ozk_prof_enable(2) (MI355X: 100 011.8 kHz for a nominal 100 MHz); 0 before that */
double ozk_prof_clock_khz(void);
int ozk_var_msm_plan(int32_t n, int32_t* window_bits, int32_t* windows);
/* 1 when the MSM of n pairs runs as 2n half-length pairs through the GLV endomorphism (the windows
 * reported above then cover 128 bits); 0 otherwise (n > 2^23 or OZK_MSM_GLV=0). */
int ozk_var_msm_glv(int32_t n);

/* Synthetic inputs for benchmarks / full-size tests (BASELINE.md config 2 generator):
 * writes n G1 bases P_i = k_i * G, k_i = splitmix64(seed + i) (k_i = 1 if that is 0), in the
 * wire-in format (affine, Z = 1).  Not part of the reference's surface. */
int ozk_gen_bases_dev(uint64_t seed, int32_t n, int32_t type, void* d_out_wire, void* stream);

/* ---------------- FixedBaseMSM ------------------------------------------
 * replaces Java_algebra_msm_FixedBaseMSM_batchMSMNativeHelper
 * (algebra_msm_FixedBaseMSM.h:13-16, .cu:1276-1384).  out: n x 192 B / n x 384 B (BE). */
int ozk_fixed_batch_msm_host(int32_t outerc, int32_t window_size, int32_t out_len,
                             int32_t inner_len, int32_t n, int32_t scalar_size,
                             const uint8_t* base, const uint8_t* scalars, int32_t bn_type,
                             int32_t task_id, uint8_t* out);
/* replaces ..._doubleBatchMSMNativeHelper (.h:21-24, .cu:1395-1491).
 * out: n x 576 B, per element G1 (3 x 64 BE) || G2 (6 x 64 BE). */
int ozk_fixed_double_batch_msm_host(int32_t outerc1, int32_t window_size1, int32_t outerc2,
                                    int32_t window_size2, int32_t out_len1, int32_t inner_len1,
                                    int32_t out_len2, int32_t inner_len2, int32_t n,
                                    const uint8_t* base_g1, const uint8_t* base_g2,
                                    const uint8_t* scalars, int32_t task_id, uint8_t* out);
/* replaces ..._fieldBatchMSMNativeHelper (.h:29-32, .cu:1500-1558).
 * in: (n+1) x 32 B LE, element n is the multiplier; out: n x 64 B BE, x_i * b mod r. */
int ozk_field_batch_mul_host(const uint8_t* in, int32_t n, int32_t task_id, uint8_t* out);

size_t ozk_fixed_batch_msm_workspace_bytes(int32_t outerc, int32_t window_size, int32_t n,
                                           int32_t bn_type);
int ozk_fixed_batch_msm_dev(int32_t outerc, int32_t window_size, int32_t n, const void* d_base,
                            const void* d_scalars, int32_t bn_type, void* d_out,
                            void* d_workspace, size_t workspace_bytes, void* stream);
int ozk_field_batch_mul_dev(const void* d_in, int32_t n, void* d_out, void* stream);
/* Compact output (no counterpart in the reference, whose natives return 64-byte
 * big-endian coordinates, algebra_msm_FixedBaseMSM.cu:783-787, i.e. 2x the bytes, and whose Java then
 * re-marshals every key element for each proof, VariableBaseMSM.java:221-228): the same points, written
 * as X|Y|Z 32-byte LITTLE-endian values — n x 96 B (G1) / n x 192 B (G2), exactly the wire-IN format of
 * the variable-base natives, so a proving key goes from the setup to the prover as it is. */
int ozk_fixed_batch_msm_compact_dev(int32_t outerc, int32_t window_size, int32_t n, const void* d_base,
                                    const void* d_scalars, int32_t bn_type, void* d_out, void* d_workspace,
                                    size_t workspace_bytes, void* stream);
int ozk_fixed_batch_msm_compact_host(int32_t outerc, int32_t window_size, int32_t n, const uint8_t* base,
                                     const uint8_t* scalars, int32_t bn_type, int32_t task_id, uint8_t* out);
/* Device-resident scalars and results, the base point given as HOST bytes (wire format, 96 / 192 B): the window
 * table — what the reference's Java side computes once per key element (getWindowTable, FixedBaseMSM.java:71-99) and
 * its native side rebuilds inside every call (algebra_msm_FixedBaseMSM.cu:851-992) — comes from a per-device cache
 * keyed by the base, so the second and later batches over one generator skip the doubling chain and the table
 * (0.65 / 2.1 ms of a 2.0 / 5.5 ms G1 / G2 call at 2^20).  The `*_host` entry points use the same cache.
 * compact != 0: the 32-byte little-endian layout of ozk_fixed_batch_msm_compact_dev.  Workspace:
 * ozk_fixed_batch_msm_workspace_bytes.  ozk_host_cache_release() frees the cached tables. */
int ozk_fixed_batch_msm_base_dev(int32_t outerc, int32_t window_size, int32_t n, const uint8_t* base_host,
                                 const void* d_scalars, int32_t bn_type, void* d_out, int32_t compact,
                                 void* d_workspace, size_t workspace_bytes, void* stream);

/* ---------------- radix-2 FFT over Fr -----------------------------------
 * replaces Java_algebra_fft_FFTAuxiliary_serialRadix2FFTNativeHelper
 * (algebra_fft_FFTAuxiliary.h:13-16, .cu:219-260) = FFTAuxiliary.serialRadix2FFT
 * (FFTAuxiliary.java:60-124).  in: n x 32 B LE (the shim flattens the List<byte[]>),
 * omega: 32 B LE, out: n x 64 B LE.  n must be a power of two (n == 1: copy). */
int ozk_fft_host(const uint8_t* in, int32_t n, const uint8_t* omega, int32_t task_id,
                 uint8_t* out);
size_t ozk_fft_workspace_bytes(int32_t n);
/* d_in: n x 32 B LE, d_out: n x 64 B LE (may not alias d_in). */
int ozk_fft_dev(const void* d_in, int32_t n, const uint8_t* omega_host32, void* d_out,
                void* d_workspace, size_t workspace_bytes, void* stream);
/* Compact form: one flat buffer in, n x 32 B LE out (the reference's native walks a
 * java.util.List<byte[]> with one JNI call per element and returns 64-byte values,
 * algebra_fft_FFTAuxiliary.cu:228-255). */
int ozk_fft_compact_host(const uint8_t* in, int32_t n, const uint8_t* omega, int32_t task_id, uint8_t* out);
int ozk_fft_compact_dev(const void* d_in, int32_t n, const uint8_t* omega_host32, void* d_out,
                        void* d_workspace, size_t workspace_bytes, void* stream);

/* ---------------- QAP witness map (the FFT path's caller) -------------
 * What R1CStoQAP.R1CStoQAPWitness (reductions/r1cs_to_qap/R1CStoQAP.java:163-230) does between the
 * constraint evaluations and the H query of the prover: 3 inverse FFTs, 3 coset FFTs, (A o B - C) / Z on
 * the coset (SerialFFT.java:86-115,158-163; FFTAuxiliary.multiplyByCoset :224-232), 1 coset inverse FFT,
 * one trailing zero — seven transforms and the pointwise stages without leaving HBM.  There is no JNI
 * native for it in the reference (Java loops over List<Fp>); INTEGRATION.md shows the optional binding.
 * A, B, C: m x 32 B LE (evaluations on the domain S, m a power of two >= 2); omega: the domain's root of
 * unity (SerialFFT.java:24-28), g: the coset shift (Fp.multiplicativeGenerator), 32 B LE each;
 * H: (m + 1) x 32 B LE coefficients (canonical).  */
/* The step before: the constraint evaluations themselves (R1CStoQAP.java:143-160,195-199 with
 * LinearCombination.evaluate, relations/objects/LinearCombination.java:39-50 — a term with variable index 0
 * contributes `one` whatever its coefficient).  A sparse matrix in CSR form resident in HBM — row_ptr: rows + 1
 * u32 offsets, index: u32 variable indices, coeff: 32-byte LE coefficients, one per term, or NULL when every
 * coefficient is one — times the assignment (32-byte LE elements): out[i] = sum over row i, rows x 32 B LE,
 * canonical.  long_rows lists the rows with more than 64 terms (n_long of them, u32): each is cut into 64
 * slices summed by one workgroup each (workspace: ozk_r1cs_evaluate_workspace_bytes(n_long)); the caller finds
 * them once per R1CS from row_ptr.  Device pointers; asynchronous on `stream`. */
size_t ozk_r1cs_evaluate_workspace_bytes(int32_t n_long);
int ozk_r1cs_evaluate_dev(const void* d_row_ptr, const void* d_index, const void* d_coeff, const void* d_assignment,
                          int32_t rows, const void* d_long_rows, int32_t n_long, void* d_out, void* d_workspace,
                          size_t workspace_bytes, void* stream);
/* ---------------- QAP instance of the setup (R1CStoQAP.R1CStoQAPRelation,
 * reductions/r1cs_to_qap/R1CStoQAP.java:37-98; SerialSetup.java:50-74,146-151) — pieces, all device-resident:
 *   ozk_qap_lagrange_dev    L_i(t) for the radix-2 domain of size m (FFTAuxiliary.serialRadix2LagrangeCoefficients,
 *                           FFTAuxiliary.java:250-302; one shared inversion per 8 coefficients instead of m inversions)
 *                           and Z(t) = t^m - 1.  t must not lie in the domain (t^m != 1: the caller checks).
 *   ozk_sparse_mat_vec_dev  out = M v for a CSR matrix in HBM: At / Bt / Ct are the TRANSPOSED constraint matrices
 *                           (input-consistency rows appended to A) times the Lagrange vector.  Same layout and
 *                           long-row list as ozk_r1cs_evaluate_dev, without its rule for variable 0.
 *   ozk_fr_powers_dev       out[i] = base^i k  (Ht, and the H query's scalars t^i Z / delta)
 *   ozk_fr_lincomb3_dev     out[i] = (ka a_i + kb b_i + c_i) kk  (the gammaABC / deltaABC scalars)
 * All vectors are n x 32-byte little-endian canonical values; 32-byte host arguments are LE canonical too. */
size_t ozk_qap_lagrange_workspace_bytes(int32_t m);
int ozk_qap_lagrange_dev(const uint8_t* t_host32, const uint8_t* omega_host32, int32_t m, void* d_out, void* d_zt,
                         void* d_workspace, size_t workspace_bytes, void* stream);
int ozk_sparse_mat_vec_dev(const void* d_row_ptr, const void* d_index, const void* d_coeff, const void* d_vec,
                           int32_t rows, const void* d_long_rows, int32_t n_long, void* d_out, void* d_workspace,
                           size_t workspace_bytes, void* stream);
size_t ozk_fr_powers_workspace_bytes(int32_t n);
int ozk_fr_powers_dev(const uint8_t* base_host32, const uint8_t* k_host32, int32_t n, void* d_out, void* d_workspace,
                      size_t workspace_bytes, void* stream);
int ozk_fr_lincomb3_dev(const void* d_a, const void* d_b, const void* d_c, int32_t n, const uint8_t* ka_host32,
                        const uint8_t* kb_host32, const uint8_t* kk_host32, void* d_out, void* d_scratch96, void* stream);
int ozk_qap_witness_host(const uint8_t* A, const uint8_t* B, const uint8_t* C, int32_t m, const uint8_t* omega,
                         const uint8_t* g, int32_t task_id, uint8_t* H);
size_t ozk_qap_witness_workspace_bytes(int32_t m);
int ozk_qap_witness_dev(const void* d_A, const void* d_B, const void* d_C, int32_t m, const uint8_t* omega_host32,
                        const uint8_t* g_host32, void* d_H, void* d_workspace, size_t workspace_bytes,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OZK_H */
