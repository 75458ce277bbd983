/* zang_hip.h -- C ABI of libzang_hip.so: zang's module paint() hot path on MI355X (gfx950).
 *
 * The reference (dbandstra/zang, Zig) has no FFI boundary: a module is a Zig struct with
 *     pub fn paint(self, span: zang.Span, outputs: [num_outputs][]f32, temps: [num_temps][]f32,
 *                  note_id_changed: bool, params: Params) void
 * (canonical form: src/modules/SineOsc.zig:8-31; the one type-erased form is
 * ModuleBase.paintFn, src/zangscript/runtime.zig:149-162).  Each `zh_<module>_paint` below
 * replaces that method for a BATCH of n independent voices (= n Zig module instances) in
 * one call; one wavefront lane renders one voice.  Argument order and meaning follow the
 * Zig signature: (self, span.start, span.end, outputs, temps, note_id_changed, params).
 *
 * Sample buffers live in device memory as images laid out [frame][voice] (voice is the
 * fastest index, so the 64 lanes of a wavefront store 256 contiguous bytes per frame).
 * A reference `[]f32` of voice v is column v of an image: element (f, v) = ptr[f*stride + v].
 *
 * Like the reference, every paint ACCUMULATES into outputs[0] (`+=`), and the caller
 * zeroes first (e.g. examples/modules.zig:220-221).  ZH_PAINT_ZERO_FIRST fuses that
 * `zang.zero(span, out)` into the paint kernel (bit-identical to zero-then-paint).
 *
 * All entry points return 0 on success, a hipError_t (> 0) when HIP failed, or a negative
 * ZH_ERR_* for bad arguments.  Work is enqueued on the context's stream and is
 * asynchronous unless stated; host arrays passed to get/set/upload/download calls are
 * synchronous.  One context per host thread; a module instance must not be painted
 * concurrently (same rule as the reference).  Nothing here allocates inside a paint (the voice
 * mixdown grows a per-context scratch buffer the first time a larger size is needed).  Destroy
 * modules and graphs before their context.
 */
#ifndef ZANG_HIP_H
#define ZANG_HIP_H

#if !defined(__HIPCC_RTC__)   /* hiprtc has no libc headers; it predefines the fixed-width types */
#include <stddef.h>
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define ZH_API __attribute__((visibility("default")))

enum { ZH_OK = 0, ZH_ERR_INVALID = -1, ZH_ERR_UNSUPPORTED = -2, ZH_ERR_NO_DEVICE = -3,
       ZH_ERR_COMM = -4,          /* librccl could not be loaded, or a rendezvous ran out of time (zh_comm_last_error says which) */
       ZH_ERR_RCCL_BASE = -100    /* an RCCL call failed: the code is ZH_ERR_RCCL_BASE - ncclResult_t */ };

/* paint flags */
enum { ZH_PAINT_ADD = 0,         /* out[i] += value          (the reference contract) */
       ZH_PAINT_ZERO_FIRST = 1,  /* zang.zero(span,out) then paint, in one kernel (basics.zig:12) */
       /* (2 = ZH_MIX_SEQUENTIAL, a zh_mixdown_voices flag) */
       ZH_PAINT_PARAMS_UNCHANGED = 4,
       ZH_PAINT_TOLERANT = 8
       /* The caller states that `params` -- scalars AND the contents of every per-voice array -- are what this module's
        * previous paint call was given.  The reference recomputes a paint's per-voice constants on every call
        * (e.g. PulseOsc.zig:87-95); a module may instead reuse the ones that call left behind.  Honoured by the
        * constant-frequency PulseOsc / TriSawOsc paints, ignored elsewhere; results are bit-identical either way.
        * A paint recorded into a graph with this flag uses the constants the module held when it was RECORDED (those of
        * the last unflagged eager paint before the capture): from then on the module stores no new constants -- later
        * unflagged paints compute theirs without keeping them, later flagged paints take the computing form.
        *
        * ZH_PAINT_TOLERANT (opt-in): the caller accepts results within 1e-5 of the signal's peak instead of the reference's bits
        * (north_star: "1e-5 relative f32", bits only for Gate and Decimator).  Honoured by
        *  - zh_filter_paint (constant or control-image cutoff / resonance) and zh_noise_filter_paint (white noise) at up to 16,384 voices,
        *    where the span is then filtered as 32-128-frame chunks at once (csrc/filter_tp.hip.h: zero-state responses, a 2 x 2
        *    transition power, then the reference's own recurrence per chunk; measured error <= 7.3e-6 of the voice's peak over 9,200 random cases,
        *    2.5-3 x faster); the noise samples and generator states are exact, the filter state carries the samples' error;
        *  - zh_nice_paint, zh_nice_paint_mix and zh_nice_paint_mix_stereo at up to 16,384 voices, spans of 128-4,096 frames: the same
        *    for the fused voice's filter (oscillator, envelope and their states exact: the envelope is walked once per voice ahead
        *    of the chunks);
        *  - zh_nice_paint_mix, zh_nice_paint_mix_stereo and zh_nice_paint_mix_stereo_batch ABOVE 16,384 voices: the exact kernel's own
        *    source compiled with multiply-adds fused (csrc/nice_mix_fma.hip: a v_fma_f32 rounds once where the reference rounds
        *    twice; 27 % fewer instructions on a kernel bound by the instructions it issues: 85 against 105 us per buffer at 131,072
        *    voices); phase counters, envelope stages and envelope clocks stay exact; measured error of a mixed sample 1e-3 of the
        *    bound (the sum of the voices' tolerances) over 24 carried buffers;
        *  - zh_noise_paint with ZH_NOISE_PINK at up to 16,384 voices: Kellett's six one-pole taps as chunks at once over exactly
        *    generated white noise (white noise itself, and the generator's state, are always exact);
        *  - zh_sineosc_paint and zh_pmosc_paint (its carrier) at any voice count: the sine of the reference's own rounded
        *    argument in f32 (csrc/zmath.hip.h zsinf_tol: 17 instructions for musl's 34, 15 of them f64; within 4e-7 of it);
        *    phase and envelope states stay exact;
        *  - zh_filtered_echoes_paint at up to 6,144 voices when the span is at most three delay lengths (input image not the
        *    output image): over a piece of <= delay_samples frames the ring's slots were all written before the piece, so the
        *    filter's input is known up front and the piece is filtered as chunks at once (csrc/delay.hip k_fe_tp_a / _b); the
        *    ring carries a paint's error into later ones (measured <= 4.5e-6 of the peak over 200 buffers, feedback 0.9; over 300 further
        *    fuzz seeds of 3-7 carried paints each, 4,096 voices, one voice reached 1.01e-5: profiles/r05/fuzz_long.txt);
        *  - zh_script_module_paint at any voice count: the generated kernel's SineOsc calls and sin() whose result reaches the
        *    output through scaling and adding alone (+ - * neg abs min max, a Filter's or Decimator's input, a delay ring written)
        *    take the f32 sine; one that reaches anything else -- another oscillator's freq or phase, a Distortion, a divisor, pow,
        *    sqrt, sin / cos -- stays exact (csrc/zscript_emit.hip decides per call when the kernel is generated).  The error is
        *    relative to the largest magnitude on the voice's signal path: where large terms cancel it is that of the terms.
        *    (The role-wave form of a generated kernel, zs_paint_pc_<name>, taken at few voices, carries no f32 sine: a tolerant
        *    paint that takes it has the exact bits.)
        * Ignored elsewhere: every other form stays bit-exact, and so does a tolerant Filter paint's first chunk.
        * NO EXCEPTION to the bound (round 5 had one): a voice whose clamped cutoff falls below 2^-9 anywhere in the paint
        * (csrc/filter_tp.hip.h kTpExactCutBelow) is not painted as chunks -- the reference's f32 accumulation of that nearly
        * constant increment drifts systematically from exact arithmetic and a chunked evaluation would depart from it by up to
        * 1.9e-5 of the peak; one lane walks that voice's frames in the reference's own order instead (bit-exact), in the same
        * launch.  Worst over tools/fuzz_tolerant.py's 10,000 seeds: 7.2e-6 of the peak (profiles/r06/fuzz_tolerant_10000.txt).
        * THE BOUND IS PER PAINT, from the state the paint starts on: every sample within 1e-5 of the voice's peak of what the
        * reference paints from that same state.  The filter state a tolerant paint leaves carries the paint's error into the next
        * one.  With damping (a resonance input below 1) that error decays, and a run whose state is carried on the GPU stays
        * inside 1e-5 against the reference carried on its own side (tests/test_gpu_tolerant.py: 200 buffers, Filter / Noise ->
        * Filter / NiceInstrument over config 3's parameter range).  In the undamped corner (resonance input >= 1 clamps the
        * damping to zero, Filter.zig:118) the filter is a lossless resonator and ANY two f32 evaluation orders of its recurrence
        * -- the reference's loop against this library's chunks, or against exact arithmetic -- drift apart like a random walk,
        * ~6e-8 * sqrt(5 * frames) of the state's amplitude (profiles/r05/tolerant_error_floor.txt: chunk start states computed
        * exactly are no closer to the reference than the f32 ones), and a phase difference of a resonator shows as a sample
        * difference that keeps growing: a carried run there departs from the reference by a few 1e-6 of the amplitude per
        * 1,024-frame buffer (tested: within 1e-5 * buffers), whatever form paints it. */ };

typedef struct zh_ctx zh_ctx;

/* Device sample image [frame][voice]; `ptr` is a DEVICE pointer (hipMalloc / torch tensor). */
typedef struct zh_buf {
    float   *ptr;
    uint32_t voices;   /* voices covered by this view                   */
    uint32_t frames;   /* frames (rows) available; spans index into it  */
    uint32_t stride;   /* floats between consecutive frames (>= voices, <= 2^26) */
    uint32_t reserved;
} zh_buf;

/* A per-voice f32 parameter: one value broadcast to every voice, or a device array[n_voices]. */
typedef struct zh_f32 { float value; uint32_t reserved; const float *per_voice; } zh_f32;
/* A per-voice bool parameter (note_on, note_id_changed): broadcast, or device uint8[n_voices]. */
typedef struct zh_bool { uint32_t value; uint32_t reserved; const uint8_t *per_voice; } zh_bool;

/* zang.ConstantOrBuffer (src/zang/constant_or_buffer.zig:4-15).  A buffer is indexed by
 * ABSOLUTE frame ([span.start..span.end], e.g. SineOsc.zig:56). */
enum { ZH_COB_CONSTANT = 0, ZH_COB_BUFFER = 1 };
typedef struct zh_cob { uint32_t tag; uint32_t reserved; zh_f32 constant; zh_buf buffer; } zh_cob;

/* zang.PaintCurve (src/zang/painter.zig:25-30); the tag is shared by all voices. */
enum { ZH_CURVE_INSTANTANEOUS = 0, ZH_CURVE_LINEAR = 1, ZH_CURVE_SQUARED = 2, ZH_CURVE_CUBED = 3 };
typedef struct zh_curve { uint32_t tag; uint32_t reserved; zh_f32 duration; } zh_curve;

/* ---------------------------------------------------------------- context, memory */
ZH_API int  zh_create(zh_ctx **out, int device);
ZH_API int  zh_destroy(zh_ctx *ctx);
ZH_API int  zh_set_stream(zh_ctx *ctx, void *hip_stream);   /* adopt an external hipStream_t; NULL = HIP's default (null) stream.
                                                                 zh_create starts on a private non-blocking stream. */
ZH_API void *zh_get_stream(zh_ctx *ctx);
ZH_API int  zh_sync(zh_ctx *ctx);
ZH_API const char *zh_error_string(int err);
ZH_API const char *zh_version(void);

ZH_API int  zh_malloc(zh_ctx *ctx, void **dev_ptr, size_t bytes);
ZH_API int  zh_free(zh_ctx *ctx, void *dev_ptr);
ZH_API int  zh_upload(zh_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes);    /* synchronous */
ZH_API int  zh_download(zh_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes);  /* synchronous */

/* out->stride >= voices: rows that would be a multiple of 64 KiB are padded by 4 KiB (HBM bank spreading); always
 * address a sample as ptr[frame * stride + voice]. */
ZH_API int  zh_buf_alloc(zh_ctx *ctx, zh_buf *out, uint32_t voices, uint32_t frames);
ZH_API int  zh_buf_free(zh_ctx *ctx, zh_buf *buf);
/* host image [voice][frame] (one contiguous reference-style []f32 per voice) <-> device [frame][voice] */
ZH_API int  zh_buf_upload_voices(zh_ctx *ctx, zh_buf dst, const float *host_voice_major, uint32_t frames);
ZH_API int  zh_buf_download_voices(zh_ctx *ctx, float *host_voice_major, zh_buf src, uint32_t frames);
/* one voice's []f32 */
ZH_API int  zh_buf_upload_voice(zh_ctx *ctx, zh_buf dst, uint32_t voice, const float *host, uint32_t frames);
ZH_API int  zh_buf_download_voice(zh_ctx *ctx, float *host, zh_buf src, uint32_t voice, uint32_t frames);

/* hipGraph capture of a launch sequence: everything enqueued on the context between begin and
 * end is recorded instead of run; zh_graph_launch replays it with one host call.  A buffer
 * loop (zero + paint + mix per 1024-frame buffer) is launch-bound at small voice counts, and
 * this removes the per-kernel host cost.  (The chunked oscillators double-buffer their phase counters and
 * flip buffers on the host at every paint; the library records which buffer a capture started from and how many
 * flips it holds, and zh_graph_launch copies the live counters over first when eager paints or other graphs
 * have left them in the other buffer -- any number of paints may be captured, and replays and eager paints mix freely.)
 * The voice mixdown's scratch must already be large enough (paint once eagerly first): it cannot grow while
 * a capture is recording (ZH_ERR_UNSUPPORTED).  Destroy a graph BEFORE the context it was captured on (zh_graph_destroy tells the
 * context that one holder of its retired scratch blocks is gone), as with the modules. */
typedef struct zh_graph zh_graph;
ZH_API int  zh_graph_begin_capture(zh_ctx *ctx);
/* The same with flags.  ZH_CAPTURE_COALESCE: the host states that while this capture records, NOTHING but zh_* calls on this
 * context is enqueued on the context's stream.  The library may then hold back paints that depend on nothing recorded before
 * them and record several as one launch: today the constant-frequency zh_pulseosc_paint / zh_trisawosc_paint (and their
 * _batch forms) flagged ZH_PAINT_PARAMS_UNCHANGED -- the reference's loop examples/write_wav.zig:58-66 painting buffer
 * after buffer with unchanged params.  Their phase counter at any frame is the counter at capture entry + frames painted
 * since * ifreq EXACTLY (`cnt +%= ifreq`, PulseOsc.zig:111, TriSawOsc.zig:115), so no paint needs the counters the previous
 * one left: consecutive paints of one module over the same span into images that do not overlap are recorded as ONE
 * launch of up to 32 buffers (exactly the launch zh_pulseosc_paint_batch makes: counters read once, written once).  When
 * something else is recorded, or the capture ends, what is held back goes out -- as two launches of half the buffers each
 * where one would leave the module's double-buffered counters on the other side (a replay then ends on the buffer it began
 * on and nothing has to be copied).  Replays give the same bits as a capture without the flag; what changes is that one
 * launch's ramp and tail are shared by the buffers.  Every other call first records what was held back, then itself, in
 * order as before.
 * Also held back: zh_nice_paint_mix_stereo (flagged ZH_PAINT_TOLERANT only from 65,536 voices on).  Its paints DO depend on each other (the voices' state), so
 * they are not reordered: consecutive ones of one instrument over one span with the same gains into different mix rows become the
 * launch zh_nice_paint_mix_stereo_batch makes, up to 8 buffers each -- the state words stay in registers from buffer to buffer and
 * the second pass runs once (105 against 108 us per buffer at 131,072 voices; same bits).  The partial-sum scratch for 8 buffers
 * is reserved by every eager stereo mixdown paint, i.e. by the eager pass a host makes before recording anyway.
 * And PIPELINED: consecutive zh_noise_filter_paint calls flagged ZH_PAINT_TOLERANT (white noise, the two-pass form, one piece each) are
 * recorded so that the second pass of paint n and the first pass of paint n + 1 are ONE launch (the first pass of a paint needs nothing
 * the second pass of the paint before it makes: it starts from the generator state the previous first pass predicted).  16 against
 * 20.5 us per buffer at 4,096 voices; the values are those of the paints recorded one after the other, except that a voice that met one
 * of Random.float's multi-draw samples (2^-41 per sample) is walked sequentially -- the reference's exact walk -- in every later paint
 * of the chain (profiles/r05/probe_overlap_nf.txt).
 * (Measured and rejected, profiles/r05/ab_capture_lanes.txt + ubench_launch_overlap.txt: the same paints as parallel graph
 * branches on 2-4 forked streams -- kernels from different queues slow each other down, 5.0-5.6 against 4.5 us per buffer.) */
enum { ZH_CAPTURE_COALESCE = 1 };
ZH_API int  zh_graph_begin_capture_flags(zh_ctx *ctx, uint32_t flags);
/* how a graph was recorded: nodes in it; paint calls that were held back while recording and the launches they became
 * (both 0 without ZH_CAPTURE_COALESCE) */
ZH_API int  zh_graph_info(const zh_graph *graph, uint32_t *nodes, uint32_t *paints_held, uint32_t *launches_of_held);
/* the kernels a replay of the graph runs, in order of first launch while recording, with their counts: "k_osc_const4[batch] x2",
 * "k_nf_tp_a x1,k_nf_tp_ba x19,k_nf_tp_b x1" ([batch]: the instantiation that paints several buffers per launch).  What
 * bench.py's roofline record names: zh_last_form after a paint CALL says nothing about a paint that was held back. */
ZH_API int  zh_graph_kernels(const zh_graph *graph, char *out, size_t n);
ZH_API int  zh_graph_end_capture(zh_ctx *ctx, zh_graph **out);
ZH_API int  zh_graph_launch(zh_ctx *ctx, zh_graph *graph);
ZH_API int  zh_graph_destroy(zh_graph *graph);

/* Which kernel form a paint takes is the library's choice, by voice count, span and arguments.  The voice-count thresholds and
 * frame-range counts behind that choice are ONE table (csrc/dispatch.hip): zh_form_count() rows, zh_form_info(i, ...) = a row's
 * name, default, current value and a line on what it selects and where it was measured.  ZH_FORMS="name=value,name=value" in
 * the environment overrides rows (read once, at the first paint; the parity tests set ZH_ENV_LIVE=1 before loading the library
 * to flip rows between paints).  zh_last_form(ctx, out, n): the kernels launched by the last entry point on this context that
 * launched any, comma-separated in launch order ("k_osc_const4", "k_nf_tp_a,k_nf_tp_b", "k_nice_mix,k_mix_pass2_wide") -- what a host, or bench.py, asks instead of guessing the form from the voice count. */
ZH_API int  zh_form_count(void);
ZH_API int  zh_form_info(int index, const char **name, long *default_value, long *current_value, const char **doc);
ZH_API int  zh_last_form(zh_ctx *ctx, char *out, size_t n);

/* timing helpers: HIP events on the context's stream (used by bench.py) */
typedef struct zh_event zh_event;
ZH_API int  zh_event_create(zh_ctx *ctx, zh_event **out);
ZH_API int  zh_event_destroy(zh_event *ev);
ZH_API int  zh_event_record(zh_ctx *ctx, zh_event *ev);
ZH_API int  zh_event_elapsed_ms(zh_event *start, zh_event *stop, float *ms);  /* synchronises on `stop` */

/* ---------------------------------------------------------------- basics.zig (src/zang/basics.zig:12-78)
 * Each op acts on frames [span_start, span_end) of every voice of `dest`. */
ZH_API int zh_zero(zh_ctx *ctx, uint32_t span_start, uint32_t span_end, zh_buf dest);                          /* :12 */
ZH_API int zh_set(zh_ctx *ctx, uint32_t span_start, uint32_t span_end, zh_buf dest, zh_f32 a);                 /* :16 */
ZH_API int zh_copy(zh_ctx *ctx, uint32_t span_start, uint32_t span_end, zh_buf dest, zh_buf src);              /* :20 */
ZH_API int zh_add(zh_ctx *ctx, uint32_t span_start, uint32_t span_end, zh_buf dest, zh_buf a, zh_buf b);       /* :24 dest += a+b */
ZH_API int zh_add_into(zh_ctx *ctx, uint32_t span_start, uint32_t span_end, zh_buf dest, zh_buf src);          /* :31 */
ZH_API int zh_add_scalar(zh_ctx *ctx, uint32_t span_start, uint32_t span_end, zh_buf dest, zh_buf a, zh_f32 b);/* :38 dest += a+b */
ZH_API int zh_add_scalar_into(zh_ctx *ctx, uint32_t span_start, uint32_t span_end, zh_buf dest, zh_f32 a);     /* :45 */
ZH_API int zh_multiply(zh_ctx *ctx, uint32_t span_start, uint32_t span_end, zh_buf dest, zh_buf a, zh_buf b);  /* :52 dest += a*b */
ZH_API int zh_multiply_with(zh_ctx *ctx, uint32_t span_start, uint32_t span_end, zh_buf dest, zh_buf a);       /* :59 */
ZH_API int zh_multiply_scalar(zh_ctx *ctx, uint32_t span_start, uint32_t span_end, zh_buf dest, zh_buf a, zh_f32 b); /* :66 dest += a*b */
ZH_API int zh_multiply_with_scalar(zh_ctx *ctx, uint32_t span_start, uint32_t span_end, zh_buf dest, zh_f32 a);      /* :73 */

/* Voice mixdown: dst[f] += sum over voices of src[f][v], f in the span -- what V successive
 * zang.addInto calls onto one mix buffer do (basics.zig:31-36; example_song.zig:382-396),
 * summed in a FIXED tree order (wave shuffle -> LDS -> block partials -> second pass) so
 * results are reproducible run to run.  `dst` is a device float[frames]. */
ZH_API int zh_mixdown_voices(zh_ctx *ctx, uint32_t span_start, uint32_t span_end, float *dst, zh_buf src, uint32_t flags);
/* flags for zh_mixdown_voices: ZH_PAINT_ZERO_FIRST, and ZH_MIX_SEQUENTIAL = add voice 0, 1, 2, ... in
 * that order in f32 exactly like successive `+=` paints onto one buffer (example_song.zig:340-346);
 * bit-faithful, meant for small voice counts (one lane per frame walks all voices). */
enum { ZH_MIX_SEQUENTIAL = 2 };

/* ---------------------------------------------------------------- multi-GPU mixdown exchange (new; no reference counterpart)
 * Voices shard across processes, one per GPU; the only data exchanged is each GPU's partial mix.
 * Besides an RCCL all-reduce issued by the host binding, the partials can be written DIRECTLY into the root GPU's
 * memory: the root allocates one slot per rank (zh_ipc_alloc) and passes the 64-byte handle to the other processes
 * by any host channel; they map it (zh_ipc_open) and hand `slot base + rank * slot_bytes` to zh_nice_paint_mix /
 * zh_mixdown_voices as the mix pointer -- the mixdown kernel's stores then travel over xGMI.  When every rank has
 * synchronised its stream (host-side barrier), the root adds the slots in rank order (zh_sum_slots): a fixed
 * order, reproducible bit for bit.  zh_free releases a zh_ipc_alloc block, zh_ipc_close a mapping. */
ZH_API int zh_ipc_alloc(zh_ctx *ctx, size_t bytes, void **dev_ptr, uint8_t *handle64 /* out: 64 bytes */);
ZH_API int zh_ipc_open(zh_ctx *ctx, const uint8_t *handle64, void **dev_ptr);
ZH_API int zh_ipc_close(zh_ctx *ctx, void *dev_ptr);
/* dst[i] (+)= ((slot_0[i] + slot_1[i]) + slot_2[i]) + ... for i < n; slot_r = slots + r * slot_stride_floats.
 * flags: ZH_PAINT_ZERO_FIRST overwrites dst. */
ZH_API int zh_sum_slots(zh_ctx *ctx, float *dst, const float *slots, uint32_t n_slots, size_t slot_stride_floats,
                        size_t n, uint32_t flags);

/* The same exchange as a COLLECTIVE: an RCCL communicator over the GPUs of the node (xGMI), one rank per process, and
 * the partial mixes summed in place on the context's stream, right behind the mixdown kernels that produced them --
 * north_star's "single RCCL reduce over xGMI for the final stereo mixdown".  It replaces the host-side accumulation of
 * the reference's buffer loop (examples/write_wav.zig:58-93: per buffer, every voice paints `+=` into the output
 * channels, then zang.mixDown; examples/example_stereo.zig:84-100 for the two channels), across GPUs.
 * librccl is opened with dlopen on first use (env ZH_RCCL_LIB overrides the search: the copy already in the process,
 * then the sibling of the loaded HIP runtime, then the ROCm installation's).
 *   rank 0:      zh_comm_unique_id(id)  -> hand the 128 bytes to every other process (any host channel)
 *   every rank:  zh_comm_create(ctx, world, rank, id, &comm)     (returns when all `world` ranks have called it, or at the limit)
 *   per batch:   zh_nice_paint_mix[_stereo] ... ; zh_allreduce_mix(comm, mix, n)  or  zh_reduce_mix(..., root)
 * `mix` is a device float[n] of this rank's GPU (e.g. [buffers][channels][frames]); the sum order is RCCL's (ring /
 * tree by size), so unlike zh_sum_slots the bits may differ between world sizes.  Calls on one communicator must be
 * issued in the same order on every rank.  The collectives may be recorded into a graph (zh_graph_begin_capture ...: RCCL
 * supports stream capture), e.g. one per buffer next to the mixdown paints; creating / destroying a communicator may not.  zh_comm_available() = 1 when librccl and its symbols were found.
 * BOUNDED RENDEZVOUS: zh_comm_create meets the other ranks in RCCL's bootstrap.  The communicator is created non-blocking
 * (ncclCommInitRankConfig, blocking = 0) and the calling thread polls ncclCommGetAsyncError; when not every rank has arrived
 * within the limit -- 180 s by default, zh_comm_set_timeout(seconds) or ZH_COMM_TIMEOUT_S in the environment change it,
 * <= 0 = wait for ever -- the half-made communicator is ended with ncclCommAbort and the call returns ZH_ERR_COMM with the
 * reason in zh_comm_last_error().  No helper thread, nothing left behind: the host may retry with a fresh id.
 * It can also return early on ONE rank without entering the bootstrap (ZH_ERR_UNSUPPORTED during a graph capture,
 * ZH_ERR_INVALID for bad arguments / out of memory, ZH_ERR_COMM when librccl is missing): the others then run into the limit.
 * A host that would rather not wait agrees over its own channel BEFORE calling it that every rank (a) sees
 * zh_comm_available() == 1, (b) is not capturing and (c) got the id -- zang_amd.sharding.Comm does exactly that with a MIN
 * all-reduce of the flags; tests/cpp/comm_host.c relies on pipe EOF.
 * AFTER CREATION: zh_comm_check(comm) surfaces ncclCommGetAsyncError -- ZH_OK while nothing is wrong (an operation still in
 * progress is not an error), ZH_ERR_RCCL_BASE - x once a peer died or a transport failed; call it between batches (bench.py
 * does once per timed region).  zh_comm_abort(comm) ends a communicator whose peers may be gone without the collective
 * hand-shake of zh_comm_destroy. */
enum { ZH_COMM_ID_BYTES = 128 };
typedef struct zh_comm zh_comm;
ZH_API int  zh_comm_available(void);
ZH_API const char *zh_comm_library(void);      /* path librccl was opened from ("" if none) */
ZH_API int  zh_comm_version(void);             /* ncclGetVersion, 0 if unavailable */
ZH_API const char *zh_comm_last_error(void);   /* text of this thread's last ZH_ERR_COMM / ZH_ERR_RCCL_BASE-x result */
ZH_API int  zh_comm_unique_id(uint8_t *id128 /* out: ZH_COMM_ID_BYTES */);
ZH_API int  zh_comm_create(zh_ctx *ctx, uint32_t world, uint32_t rank, const uint8_t *id128, zh_comm **out);
ZH_API int  zh_comm_set_timeout(double seconds);   /* process-wide limit of the rendezvous and of calls RCCL answers "in progress" */
ZH_API int  zh_comm_check(zh_comm *comm);
ZH_API int  zh_comm_destroy(zh_comm *comm);    /* synchronises the context's stream first */
ZH_API int  zh_comm_abort(zh_comm *comm);      /* ncclCommAbort: no hand-shake, nothing synchronised */
ZH_API int  zh_comm_world(const zh_comm *comm);
ZH_API int  zh_comm_rank(const zh_comm *comm);
ZH_API int  zh_allreduce_mix(zh_comm *comm, float *mix, size_t n);                 /* every rank ends with the sum */
ZH_API int  zh_reduce_mix(zh_comm *comm, float *mix, size_t n, uint32_t root);     /* only `root` ends with the sum */

/* zang.mixDown (src/zang/mixdown.zig:8-86): f32 mix buffer -> interleaved signed 8 / 16-bit LE PCM with
 * clamping.  `dst` (device bytes, n * bytes_per_sample * num_channels) and `mix` (device float[n]). */
enum { ZH_AUDIO_SIGNED8 = 0, ZH_AUDIO_SIGNED16_LSB = 1 };                                             /* :3-6 */
ZH_API int zh_mix_down(zh_ctx *ctx, uint8_t *dst, const float *mix, uint32_t n, uint32_t audio_format,
                       uint32_t num_channels, uint32_t channel_index, float vol);

/* ---------------------------------------------------------------- SineOsc (src/modules/SineOsc.zig) */
typedef struct zh_sineosc zh_sineosc;
typedef struct zh_sineosc_params { float sample_rate; uint32_t reserved; zh_cob freq; zh_cob phase; } zh_sineosc_params; /* :10-14 */
typedef struct zh_sineosc_state { float t; } zh_sineosc_state;                                        /* :16 */
ZH_API int zh_sineosc_create(zh_ctx *ctx, uint32_t n_voices, zh_sineosc **out);                       /* n x init() :18-22 */
ZH_API int zh_sineosc_destroy(zh_sineosc *m);
ZH_API int zh_sineosc_get_state(zh_sineosc *m, zh_sineosc_state *host);
ZH_API int zh_sineosc_set_state(zh_sineosc *m, const zh_sineosc_state *host);
ZH_API int zh_sineosc_paint(zh_sineosc *m, uint32_t span_start, uint32_t span_end, const zh_buf *outputs /*[1]*/,
                            const zh_buf *temps /*[0]*/, zh_bool note_id_changed,
                            const zh_sineosc_params *params, uint32_t flags);                         /* :24-87 */

/* ---------------------------------------------------------------- PulseOsc (src/modules/PulseOsc.zig) */
typedef struct zh_pulseosc zh_pulseosc;
typedef struct zh_pulseosc_params { float sample_rate; uint32_t reserved; zh_cob freq; zh_f32 color; } zh_pulseosc_params; /* :30-34 */
typedef struct zh_pulseosc_state { uint32_t cnt; } zh_pulseosc_state;                                 /* :36 */
ZH_API int zh_pulseosc_create(zh_ctx *ctx, uint32_t n_voices, zh_pulseosc **out);
ZH_API int zh_pulseosc_destroy(zh_pulseosc *m);
ZH_API int zh_pulseosc_get_state(zh_pulseosc *m, zh_pulseosc_state *host);
ZH_API int zh_pulseosc_set_state(zh_pulseosc *m, const zh_pulseosc_state *host);
ZH_API int zh_pulseosc_paint(zh_pulseosc *m, uint32_t span_start, uint32_t span_end, const zh_buf *outputs,
                             const zh_buf *temps, zh_bool note_id_changed,
                             const zh_pulseosc_params *params, uint32_t flags);                       /* :44-157 */
/* n_buffers consecutive paint calls with the same span and params, buffer b into outputs[b] (what a host loop over
 * 1024-frame buffers does, examples/write_wav.zig:58-66): with a constant frequency the phase counter of any frame is
 * cnt + frames_before * ifreq exactly (:111), so the whole batch is one launch.  State afterwards = after the last call. */
ZH_API int zh_pulseosc_paint_batch(zh_pulseosc *m, uint32_t span_start, uint32_t span_end, const zh_buf *outputs /*[n_buffers]*/,
                                   uint32_t n_buffers, const zh_pulseosc_params *params, uint32_t flags);

/* ---------------------------------------------------------------- TriSawOsc (src/modules/TriSawOsc.zig) */
typedef struct zh_trisawosc zh_trisawosc;
typedef struct zh_trisawosc_params { float sample_rate; uint32_t reserved; zh_cob freq; zh_f32 color; } zh_trisawosc_params; /* :30-34 */
typedef struct zh_trisawosc_state { uint32_t cnt; float t; } zh_trisawosc_state;                      /* :36-37 */
ZH_API int zh_trisawosc_create(zh_ctx *ctx, uint32_t n_voices, zh_trisawosc **out);
ZH_API int zh_trisawosc_destroy(zh_trisawosc *m);
ZH_API int zh_trisawosc_get_state(zh_trisawosc *m, zh_trisawosc_state *host);
ZH_API int zh_trisawosc_set_state(zh_trisawosc *m, const zh_trisawosc_state *host);
ZH_API int zh_trisawosc_paint(zh_trisawosc *m, uint32_t span_start, uint32_t span_end, const zh_buf *outputs,
                              const zh_buf *temps, zh_bool note_id_changed,
                              const zh_trisawosc_params *params, uint32_t flags);                     /* :46-156 */
ZH_API int zh_trisawosc_paint_batch(zh_trisawosc *m, uint32_t span_start, uint32_t span_end, const zh_buf *outputs /*[n_buffers]*/,
                                    uint32_t n_buffers, const zh_trisawosc_params *params, uint32_t flags);   /* see zh_pulseosc_paint_batch */

/* ---------------------------------------------------------------- Noise (src/modules/Noise.zig) */
typedef struct zh_noise zh_noise;
enum { ZH_NOISE_WHITE = 0, ZH_NOISE_PINK = 1 };                                                       /* :11-14 */
typedef struct zh_noise_params { uint32_t color; } zh_noise_params;                                   /* :18-20 */
typedef struct zh_noise_state { uint64_t r[4]; float b[7]; uint32_t reserved; } zh_noise_state;       /* :22-23 */
/* Voice v is seeded like the (first_seed + v)-th Noise.init() of a process (:9,:26): pass the
 * voice's GLOBAL index so that sharded and unsharded renders agree. */
ZH_API int zh_noise_create(zh_ctx *ctx, uint32_t n_voices, uint64_t first_seed, zh_noise **out);
ZH_API int zh_noise_destroy(zh_noise *m);
ZH_API int zh_noise_get_state(zh_noise *m, zh_noise_state *host);
ZH_API int zh_noise_set_state(zh_noise *m, const zh_noise_state *host);
ZH_API int zh_noise_paint(zh_noise *m, uint32_t span_start, uint32_t span_end, const zh_buf *outputs,
                          const zh_buf *temps, zh_bool note_id_changed,
                          const zh_noise_params *params, uint32_t flags);                             /* :34-72 */

/* Host-only self-test of the xoshiro256++ jump tables behind the frame-range form of white noise (few voices: a span is
 * painted as many frame ranges at once, each from a state jumped T^(32 j) ahead; csrc/noise_jump.hip): for n_states
 * random states, table j applied to the state must equal 32 (j + 1) sequential transitions, j = 0..62.  Returns the
 * number of mismatching state words (0 = pass).  Needs no device. */
ZH_API int zh_selftest_noise_jump(uint64_t seed, uint32_t n_states);

/* ---------------------------------------------------------------- Envelope (src/modules/Envelope.zig, src/zang/painter.zig) */
typedef struct zh_envelope zh_envelope;
enum { ZH_ENV_IDLE = 0, ZH_ENV_ATTACK, ZH_ENV_DECAY, ZH_ENV_SUSTAIN, ZH_ENV_RELEASE };                /* :15-21 */
typedef struct zh_envelope_params {                                                                   /* :6-13 */
    float sample_rate; uint32_t reserved;
    zh_curve attack, decay, release;
    zh_f32 sustain_volume;
    zh_bool note_on;
} zh_envelope_params;
typedef struct zh_envelope_state { uint32_t state; float t, last_value, start; } zh_envelope_state;   /* :23-24, painter.zig:33-36 */
ZH_API int zh_envelope_create(zh_ctx *ctx, uint32_t n_voices, zh_envelope **out);
ZH_API int zh_envelope_destroy(zh_envelope *m);
ZH_API int zh_envelope_get_state(zh_envelope *m, zh_envelope_state *host);
ZH_API int zh_envelope_set_state(zh_envelope *m, const zh_envelope_state *host);
ZH_API int zh_envelope_paint(zh_envelope *m, uint32_t span_start, uint32_t span_end, const zh_buf *outputs,
                             const zh_buf *temps, zh_bool note_id_changed,
                             const zh_envelope_params *params, uint32_t flags);                       /* :92-109 */

/* ---------------------------------------------------------------- Gate (src/modules/Gate.zig) -- stateless */
typedef struct zh_gate zh_gate;
typedef struct zh_gate_params { zh_bool note_on; } zh_gate_params;                                    /* :7-9 */
ZH_API int zh_gate_create(zh_ctx *ctx, uint32_t n_voices, zh_gate **out);
ZH_API int zh_gate_destroy(zh_gate *m);
ZH_API int zh_gate_paint(zh_gate *m, uint32_t span_start, uint32_t span_end, const zh_buf *outputs,
                         const zh_buf *temps, zh_bool note_id_changed,
                         const zh_gate_params *params, uint32_t flags);                               /* :15-30 */

/* ---------------------------------------------------------------- Filter (src/modules/Filter.zig) */
typedef struct zh_filter zh_filter;
enum { ZH_FILTER_BYPASS = 0, ZH_FILTER_LOW_PASS, ZH_FILTER_BAND_PASS, ZH_FILTER_HIGH_PASS,
       ZH_FILTER_NOTCH, ZH_FILTER_ALL_PASS };                                                         /* :10-17 */
typedef struct zh_filter_params { zh_buf input; uint32_t type; uint32_t reserved; zh_cob cutoff; zh_cob res; } zh_filter_params; /* :27-32 */
typedef struct zh_filter_state { float l, b; } zh_filter_state;                                       /* :34-35 */
ZH_API int zh_filter_create(zh_ctx *ctx, uint32_t n_voices, zh_filter **out);
ZH_API int zh_filter_destroy(zh_filter *m);
ZH_API int zh_filter_get_state(zh_filter *m, zh_filter_state *host);
ZH_API int zh_filter_set_state(zh_filter *m, const zh_filter_state *host);
ZH_API int zh_filter_paint(zh_filter *m, uint32_t span_start, uint32_t span_end, const zh_buf *outputs,
                           const zh_buf *temps, zh_bool note_id_changed,
                           const zh_filter_params *params, uint32_t flags);                           /* :44-151 */
/* Filter.cutoffFromFrequency (:20-23) for n values, on the device (host scalar in the reference). */
ZH_API int zh_filter_cutoff_from_frequency(zh_ctx *ctx, uint32_t n, float *cutoff_out_dev,
                                           const float *frequency_dev, float sample_rate);

/* std.math.pow(f32, x, y) for finite x > 0, elementwise on the device (out, x, y: device float[n]).
 * Host callers of the paint path compute scalars with it: note frequencies a4 * pow(2, semitones/12)
 * (examples/common/songparse1.zig:61-62), Distortion's gain1 (Distortion.zig:41). */
ZH_API int zh_pow(zh_ctx *ctx, uint32_t n, float *out, const float *x, const float *y);
/* std.math.sin / std.math.cos (f32), elementwise on the device for any float (inf and nan included): the
 * functions SineOsc (SineOsc.zig:4-6), PMOscInstrument and Filter.cutoffFromFrequency (Filter.zig:21) are
 * built on, exposed so that a host can compute its scalars with the same bits (out, x: device float[n]). */
ZH_API int zh_sin(zh_ctx *ctx, uint32_t n, float *out, const float *x);
ZH_API int zh_cos(zh_ctx *ctx, uint32_t n, float *out, const float *x);
/* std.math.atan (f32), the function of Distortion's overdrive (Distortion.zig:45, 50), same contract. */
ZH_API int zh_atan(zh_ctx *ctx, uint32_t n, float *out, const float *x);

/* ---------------------------------------------------------------- Sampler (src/modules/Sampler.zig) */
typedef struct zh_sampler zh_sampler;
enum { ZH_SAMPLE_U8 = 0, ZH_SAMPLE_S16_LSB, ZH_SAMPLE_S24_LSB, ZH_SAMPLE_S32_LSB };                   /* :9-14 */
typedef struct zh_sample {                                                                            /* :16-21 */
    uint64_t num_channels; uint64_t sample_rate; uint32_t format; uint32_t reserved;
    const uint8_t *data;   /* DEVICE pointer to the PCM bytes, shared read-only by all voices */
    uint64_t data_len;     /* bytes */
} zh_sample;
typedef struct zh_sampler_params {                                                                    /* :62-67 */
    zh_f32 sample_rate;    /* per-voice: example_sampler.zig varies it per note to change pitch */
    zh_sample sample; uint64_t channel; uint32_t loop; uint32_t reserved;
} zh_sampler_params;
typedef struct zh_sampler_state { float t; } zh_sampler_state;                                        /* :69 */
ZH_API int zh_sampler_create(zh_ctx *ctx, uint32_t n_voices, zh_sampler **out);
ZH_API int zh_sampler_destroy(zh_sampler *m);
ZH_API int zh_sampler_get_state(zh_sampler *m, zh_sampler_state *host);
ZH_API int zh_sampler_set_state(zh_sampler *m, const zh_sampler_state *host);
ZH_API int zh_sampler_paint(zh_sampler *m, uint32_t span_start, uint32_t span_end, const zh_buf *outputs,
                            const zh_buf *temps, zh_bool note_id_changed,
                            const zh_sampler_params *params, uint32_t flags);                         /* :77-136 */

/* ---------------------------------------------------------------- Decimator (src/modules/Decimator.zig) */
typedef struct zh_decimator zh_decimator;
typedef struct zh_decimator_params { float sample_rate; uint32_t reserved; zh_buf input; zh_f32 fake_sample_rate; } zh_decimator_params; /* :5-9 */
typedef struct zh_decimator_state { float dval, dcount; } zh_decimator_state;                         /* :11-12 */
ZH_API int zh_decimator_create(zh_ctx *ctx, uint32_t n_voices, zh_decimator **out);
ZH_API int zh_decimator_destroy(zh_decimator *m);
ZH_API int zh_decimator_get_state(zh_decimator *m, zh_decimator_state *host);
ZH_API int zh_decimator_set_state(zh_decimator *m, const zh_decimator_state *host);
ZH_API int zh_decimator_paint(zh_decimator *m, uint32_t span_start, uint32_t span_end, const zh_buf *outputs,
                              const zh_buf *temps, zh_bool note_id_changed,
                              const zh_decimator_params *params, uint32_t flags);                     /* :21-57 */

/* ---------------------------------------------------------------- Distortion (src/modules/Distortion.zig) -- stateless */
typedef struct zh_distortion zh_distortion;
enum { ZH_DISTORTION_OVERDRIVE = 0, ZH_DISTORTION_CLIP = 1 };                                         /* :8-11 */
typedef struct zh_distortion_params { zh_buf input; uint32_t type; uint32_t reserved; zh_f32 ingain, outgain, offset; } zh_distortion_params; /* :15-21 */
ZH_API int zh_distortion_create(zh_ctx *ctx, uint32_t n_voices, zh_distortion **out);
ZH_API int zh_distortion_destroy(zh_distortion *m);
ZH_API int zh_distortion_paint(zh_distortion *m, uint32_t span_start, uint32_t span_end, const zh_buf *outputs,
                               const zh_buf *temps, zh_bool note_id_changed,
                               const zh_distortion_params *params, uint32_t flags);                   /* :27-66 */

/* ---------------------------------------------------------------- Curve (src/modules/Curve.zig)
 * (entry points are named zh_curve_module_* because zh_curve is zang.PaintCurve) */
typedef struct zh_curve_module zh_curve_module;
enum { ZH_CURVE_FN_LINEAR = 0, ZH_CURVE_FN_SMOOTHSTEP = 1 };                                          /* :4-7 */
typedef struct zh_curve_node { float value, t; } zh_curve_node;                                       /* zang.CurveNode, src/zang/curve.zig:3-6 */
typedef struct zh_curve_module_params {                                                               /* :29-33 */
    float sample_rate; uint32_t function;
    const zh_curve_node *curve;   /* DEVICE array shared by all voices (must not be mutated while painting, :37-38) */
    uint64_t curve_len;
} zh_curve_module_params;
typedef struct zh_curve_module_state {                                                                /* :36-41 */
    float t; uint32_t current_song_note; int32_t current_song_note_offset; uint32_t next_song_note;
} zh_curve_module_state;
ZH_API int zh_curve_module_create(zh_ctx *ctx, uint32_t n_voices, zh_curve_module **out);
ZH_API int zh_curve_module_destroy(zh_curve_module *m);
ZH_API int zh_curve_module_get_state(zh_curve_module *m, zh_curve_module_state *host);
ZH_API int zh_curve_module_set_state(zh_curve_module *m, const zh_curve_module_state *host);
ZH_API int zh_curve_module_paint(zh_curve_module *m, uint32_t span_start, uint32_t span_end, const zh_buf *outputs,
                                 const zh_buf *temps, zh_bool note_id_changed,
                                 const zh_curve_module_params *params, uint32_t flags);               /* :56-128 */

/* ---------------------------------------------------------------- Cycle (src/modules/Cycle.zig) */
typedef struct zh_cycle zh_cycle;
typedef struct zh_cycle_params { float sample_rate; uint32_t reserved; zh_cob speed; } zh_cycle_params;          /* :9-12 */
typedef struct zh_cycle_state { float t; } zh_cycle_state;                                            /* :14 */
ZH_API int zh_cycle_create(zh_ctx *ctx, uint32_t n_voices, zh_cycle **out);
ZH_API int zh_cycle_destroy(zh_cycle *m);
ZH_API int zh_cycle_get_state(zh_cycle *m, zh_cycle_state *host);
ZH_API int zh_cycle_set_state(zh_cycle *m, const zh_cycle_state *host);
ZH_API int zh_cycle_paint(zh_cycle *m, uint32_t span_start, uint32_t span_end, const zh_buf *outputs,
                          const zh_buf *temps, zh_bool note_id_changed,
                          const zh_cycle_params *params, uint32_t flags);                             /* :22-59 */

/* ---------------------------------------------------------------- Portamento (src/modules/Portamento.zig) */
typedef struct zh_portamento zh_portamento;
typedef struct zh_portamento_params {                                                                 /* :5-11 */
    float sample_rate; uint32_t reserved; zh_curve curve; zh_f32 goal; zh_bool note_on; zh_bool prev_note_on;
} zh_portamento_params;
typedef struct zh_portamento_state { float t, last_value, start; } zh_portamento_state;               /* :13, painter.zig:33-36 */
ZH_API int zh_portamento_create(zh_ctx *ctx, uint32_t n_voices, zh_portamento **out);
ZH_API int zh_portamento_destroy(zh_portamento *m);
ZH_API int zh_portamento_get_state(zh_portamento *m, zh_portamento_state *host);
ZH_API int zh_portamento_set_state(zh_portamento *m, const zh_portamento_state *host);
ZH_API int zh_portamento_paint(zh_portamento *m, uint32_t span_start, uint32_t span_end, const zh_buf *outputs,
                               const zh_buf *temps, zh_bool note_id_changed,
                               const zh_portamento_params *params, uint32_t flags);                   /* :21-48 */

/* ---------------------------------------------------------------- NiceInstrument (examples/modules.zig:189-248)
 * PulseOsc -> x0.5 -> Filter(low_pass, cutoffFromFrequency(8*freq), res 0.7) -> Envelope(cubed
 * .01/.1/.5, sustain .8) -> out += env*flt, as ONE kernel: the two temps never touch HBM. */
typedef struct zh_nice zh_nice;
typedef struct zh_nice_params { float sample_rate; uint32_t reserved; zh_f32 freq; zh_bool note_on; } zh_nice_params; /* :192-196 */
typedef struct zh_nice_state { zh_pulseosc_state osc; zh_filter_state flt; zh_envelope_state env; } zh_nice_state;    /* :198-201 */
ZH_API int zh_nice_create(zh_ctx *ctx, uint32_t n_voices, zh_f32 color, zh_nice **out);               /* init(color) :203-210 */
ZH_API int zh_nice_destroy(zh_nice *m);
ZH_API int zh_nice_get_state(zh_nice *m, zh_nice_state *host);
ZH_API int zh_nice_set_state(zh_nice *m, const zh_nice_state *host);
ZH_API int zh_nice_paint(zh_nice *m, uint32_t span_start, uint32_t span_end, const zh_buf *outputs,
                         const zh_buf *temps /*[2], unused by the fused kernel; may be NULL*/,
                         zh_bool note_id_changed, const zh_nice_params *params, uint32_t flags);      /* :212-247 */
/* The same chain, then the voice mixdown, without materialising per-voice output:
 * partial[f] (+)= sum_v voice_v[f].  `mix` is a device float[frames]. */
ZH_API int zh_nice_paint_mix(zh_nice *m, uint32_t span_start, uint32_t span_end, float *mix,
                             zh_bool note_id_changed, const zh_nice_params *params, uint32_t flags);
/* Two output channels (a module with num_outputs = 2, examples/example_stereo.zig:42-43): every voice is added to each
 * channel scaled by that voice's channel gain, `outputs[c] += voice * pan_c` (example_stereo.zig:92-98, zang.multiply:
 * product rounded to f32, then added), and each channel is mixed down over the voices:
 * mix_c[f] (+)= sum_v voice_v[f] * gain_c[v].  mix_left / mix_right are device float[frames] (zh_mix_down then
 * interleaves them, examples/write_wav.zig:71-78). */
ZH_API int zh_nice_paint_mix_stereo(zh_nice *m, uint32_t span_start, uint32_t span_end, float *mix_left, float *mix_right,
                                    zh_f32 gain_left, zh_f32 gain_right, zh_bool note_id_changed,
                                    const zh_nice_params *params, uint32_t flags);

/* n_buffers consecutive zh_nice_paint_mix_stereo calls -- the host's loop over 1024-frame buffers (examples/write_wav.zig:58-93)
 * of an offline render that knows its params ahead -- as ONE launch: buffer b paints [span_start, span_end) with params[b] and
 * note_id_changed[b] into mix_left[b] / mix_right[b] (host arrays of n_buffers device pointers / structs); the voices' state
 * stays in registers from buffer to buffer and the second mixdown pass runs once for all buffers.  Bit-identical to n_buffers
 * single calls; the sample rate must be the same in every params[b] (ZH_ERR_UNSUPPORTED otherwise). */
/* Scratch: the fused mixdown keeps per-context partial rows in HBM -- frames (rounded up to 8) x rows x channels x 4 bytes per
 * buffer, rows = one per 256 voices from 65,536 voices up and one per 64 voices below (ZH_NICE_MIX_WG_MIN): 4 MiB per stereo
 * buffer of 1,024 frames at 131,072 voices, 32 MiB at 1,048,576; a batch reserves n_buffers times that (512 MiB for 16 buffers at
 * 1 Mi voices).  The block only grows, on the first call that needs more (never inside a capture: ZH_ERR_UNSUPPORTED -- make one
 * eager call of the size first); an outgrown block is freed at once unless a captured graph of the context is alive, then when
 * the last one is destroyed. */
enum { ZH_MIX_MAX_BATCH = 16 };
ZH_API int zh_nice_paint_mix_stereo_batch(zh_nice *m, uint32_t span_start, uint32_t span_end, uint32_t n_buffers,
                                          float *const *mix_left, float *const *mix_right, zh_f32 gain_left, zh_f32 gain_right,
                                          const zh_bool *note_id_changed, const zh_nice_params *params, uint32_t flags);

/* Per-voice span table: the output of NoteTracker -> PolyphonyDispatcher -> Trigger for one buffer
 * (examples/example_song.zig:326-349), i.e. for every voice up to `max_spans` sub-spans, ascending and
 * non-overlapping, each with the note's params and note_id_changed.  One launch then performs, per voice,
 * the same sequence of paint(sub_span, ..., note_id_changed, params) calls the reference's Trigger loop
 * makes (per-call prologue/epilogue included).  With more than 64 voices a lane owns a voice and the wave walks
 * segments between sub-span boundaries; with up to 64 voices a WAVE owns a voice and its lanes are 64 consecutive
 * frames (only what truly carries state from frame to frame -- phase and envelope clocks, the filter core -- is
 * walked frame by frame; oscillator, envelope curve and sines are evaluated for 64 frames at once) -- the same
 * values either way.  Arrays are device memory laid out [span_index][voice]; zh_poly_voice_schedule fills this layout. */
typedef struct zh_span_table {
    uint32_t max_spans, reserved;
    const uint32_t *count;                       /* [n_voices]            */
    const uint32_t *start, *end;                 /* [max_spans][n_voices] */
    const float    *freq;                        /* [max_spans][n_voices] */
    const uint8_t  *note_on, *note_id_changed;   /* [max_spans][n_voices] */
} zh_span_table;
ZH_API int zh_nice_paint_spans(zh_nice *m, uint32_t span_start, uint32_t span_end, const zh_buf *outputs,
                               const zh_buf *temps, float sample_rate, const zh_span_table *table, uint32_t flags);

/* ---------------------------------------------------------------- Delay(n) and its composites
 * zang.Delay(delay_samples) (src/zang/delay.zig:7-91) is a per-voice ring buffer; on the device the rings of
 * n voices form one image [delay_sample][voice] in HBM.  Two modules use it:
 *   zh_delay           = SimpleDelay   (examples/modules.zig:341-386): out += ring; ring = input
 *   zh_filtered_echoes = FilteredEchoes (examples/modules.zig:390-461): feedback*ring + input -> low-pass -> out, ring
 * The reference moves data in chunks of <= delay_samples (read, then write); reading a slot always precedes
 * writing it, so the per-sample form used by the kernels is equivalent.  StereoEchoes (:463-525) is the host-level
 * composition addInto / zh_delay / zh_filtered_echoes / zh_delay. */
typedef struct zh_delay zh_delay;
typedef struct zh_delay_params { zh_buf input; } zh_delay_params;                                     /* :345-347 */
ZH_API int zh_delay_create(zh_ctx *ctx, uint32_t n_voices, uint32_t delay_samples, zh_delay **out);   /* init(): ring zeros, index 0 */
ZH_API int zh_delay_destroy(zh_delay *m);
ZH_API int zh_delay_reset(zh_delay *m);                                                               /* delay.zig:19-22 */
/* state: host float[n_voices][delay_samples] (one ring per voice) and uint32 index[n_voices] */
ZH_API int zh_delay_get_state(zh_delay *m, float *rings_voice_major, uint32_t *index);
ZH_API int zh_delay_set_state(zh_delay *m, const float *rings_voice_major, const uint32_t *index);
ZH_API int zh_delay_paint(zh_delay *m, uint32_t span_start, uint32_t span_end, const zh_buf *outputs,
                          const zh_buf *temps, zh_bool note_id_changed, const zh_delay_params *params, uint32_t flags);

typedef struct zh_filtered_echoes zh_filtered_echoes;
typedef struct zh_filtered_echoes_params { zh_buf input; zh_f32 feedback_volume; zh_f32 cutoff; } zh_filtered_echoes_params; /* :394-398 */
ZH_API int zh_filtered_echoes_create(zh_ctx *ctx, uint32_t n_voices, uint32_t delay_samples, zh_filtered_echoes **out);
ZH_API int zh_filtered_echoes_destroy(zh_filtered_echoes *m);
ZH_API int zh_filtered_echoes_reset(zh_filtered_echoes *m);                                           /* :407-409: the delay only */
ZH_API int zh_filtered_echoes_get_state(zh_filtered_echoes *m, float *rings_voice_major, uint32_t *index, zh_filter_state *filter);
ZH_API int zh_filtered_echoes_set_state(zh_filtered_echoes *m, const float *rings_voice_major, const uint32_t *index, const zh_filter_state *filter);
ZH_API int zh_filtered_echoes_paint(zh_filtered_echoes *m, uint32_t span_start, uint32_t span_end, const zh_buf *outputs,
                                    const zh_buf *temps /*[2], unused; may be NULL*/, zh_bool note_id_changed,
                                    const zh_filtered_echoes_params *params, uint32_t flags);

/* ---------------------------------------------------------------- Noise -> Filter voice (examples/example_stereo.zig:71-82)
 * zero(temp); Noise.paint(temp); Filter.paint(out, input = temp, type, cutoff, res) as ONE kernel: the
 * temp image stays in registers (BASELINE config 3, fused variant).  Bit-identical to the two separate
 * paints.  cutoff / res are per-voice constants here. */
typedef struct zh_noise_filter zh_noise_filter;
typedef struct zh_noise_filter_params { uint32_t color; uint32_t type; zh_f32 cutoff; zh_f32 res; } zh_noise_filter_params;
typedef struct zh_noise_filter_state { zh_noise_state noise; zh_filter_state flt; } zh_noise_filter_state;
ZH_API int zh_noise_filter_create(zh_ctx *ctx, uint32_t n_voices, uint64_t first_seed, zh_noise_filter **out);
ZH_API int zh_noise_filter_destroy(zh_noise_filter *m);
ZH_API int zh_noise_filter_get_state(zh_noise_filter *m, zh_noise_filter_state *host);
ZH_API int zh_noise_filter_set_state(zh_noise_filter *m, const zh_noise_filter_state *host);
ZH_API int zh_noise_filter_paint(zh_noise_filter *m, uint32_t span_start, uint32_t span_end, const zh_buf *outputs,
                                 const zh_buf *temps /*[1], unused; may be NULL*/, zh_bool note_id_changed,
                                 const zh_noise_filter_params *params, uint32_t flags);

/* ---------------------------------------------------------------- PMOscInstrument (examples/modules.zig:6-128) */
typedef struct zh_pmosc zh_pmosc;
typedef struct zh_pmosc_params { float sample_rate; uint32_t reserved; zh_f32 freq; zh_bool note_on; } zh_pmosc_params; /* :83-87 */
typedef struct zh_pmosc_state { zh_sineosc_state carrier, modulator; zh_envelope_state env; } zh_pmosc_state;          /* :24-25, :89-91 */
ZH_API int zh_pmosc_create(zh_ctx *ctx, uint32_t n_voices, zh_f32 release_duration, zh_pmosc **out);  /* init(release_duration) :93-99 */
ZH_API int zh_pmosc_destroy(zh_pmosc *m);
ZH_API int zh_pmosc_get_state(zh_pmosc *m, zh_pmosc_state *host);
ZH_API int zh_pmosc_set_state(zh_pmosc *m, const zh_pmosc_state *host);
ZH_API int zh_pmosc_paint(zh_pmosc *m, uint32_t span_start, uint32_t span_end, const zh_buf *outputs,
                          const zh_buf *temps /*[3], unused; may be NULL*/,
                          zh_bool note_id_changed, const zh_pmosc_params *params, uint32_t flags);    /* :101-127 */
ZH_API int zh_pmosc_paint_spans(zh_pmosc *m, uint32_t span_start, uint32_t span_end, const zh_buf *outputs,
                                const zh_buf *temps, float sample_rate, const zh_span_table *table, uint32_t flags);

/* ---------------------------------------------------------------- zangscript modules
 * The reference compiles its module DSL to Zig source (tools/zangc.zig -> src/zangscript/codegen_zig.zig) that is
 * then built into the host program.  Here the same front-end (python -m zang_amd.zangc, zang_amd/zangscript/)
 * emits HIP source with ONE fused lane-per-voice kernel per exported module -- every temp buffer of the
 * generated Zig paint() becomes a per-frame register -- and this loader compiles it for gfx950 with hiprtc
 * and runs it.  The call shape stays the module contract of SineOsc.zig:8-31: paint(span, outputs, temps,
 * note_id_changed, params), with the script module's Params (codegen_zig.zig:520-527) passed as an array of
 * zh_script_param in declaration order (index 0 is the implicit `sample_rate`, parse.zig:330-331). */
enum { ZH_SP_CONSTANT = 0,   /* f32: `f`, or `pf` per voice                                         */
       ZH_SP_BOOLEAN = 1,    /* bool: `u`, or `pb` per voice                                        */
       ZH_SP_COB = 2,        /* zang.ConstantOrBuffer: is_buffer ? image pf/stride : constant f|pf   */
       ZH_SP_BUFFER = 3,     /* []const f32 ("waveform"): image pf/stride                           */
       ZH_SP_ENUM = 4,       /* one_of: `u` = index of the value in the enum's declaration order,
                                `f` = its f32 payload if it has one (PaintCurve durations)          */
       ZH_SP_CURVE = 5 };    /* []const zang.CurveNode: pf = DEVICE zh_curve_node array, u = count  */
typedef struct zh_script_param {
    uint32_t kind, u;
    float f;
    uint32_t is_buffer;
    const float *pf;
    const uint8_t *pb;
    uint32_t stride, reserved;
} zh_script_param;
#define ZH_SCRIPT_MAX_PARAMS 16

typedef struct zh_script zh_script;                 /* one compiled + loaded script (a hipModule) */
typedef struct zh_script_module zh_script_module;   /* n_voices instances of one exported module */
/* Compile only (no GPU needed): returns a malloc'ed gfx950 code object, or the compiler log in `log`. */
ZH_API int zh_script_compile(const char *hip_source, void **code_out, size_t *code_size_out, char *log, size_t log_cap);
ZH_API void zh_script_free_code(void *code);
ZH_API int zh_script_load(zh_ctx *ctx, const char *hip_source, zh_script **out, char *log, size_t log_cap);
/* ... or from the code object zh_script_compile returned earlier -- compiled once, offline (hiprtc needs no GPU), like the reference
 * compiles a script to Zig before the program is built (examples/example_script.zig:6-8): no hiprtc at run time (1-4 s per module). */
ZH_API int zh_script_load_code(zh_ctx *ctx, const void *code, size_t code_size, zh_script **out);
ZH_API int zh_script_destroy(zh_script *s);
/* `name` = the exported module's global name; `state_words` = 32-bit state words per voice as reported by the
 * front-end; Noise fields are seeded first_seed + voice * n_noise_fields + k (Noise.zig:25-29: one counter tick
 * per init(), instances created voice by voice). */
ZH_API int zh_script_module_create(zh_script *s, const char *name, uint32_t n_voices, uint32_t state_words,
                                   uint64_t first_seed, zh_script_module **out);
ZH_API int zh_script_module_destroy(zh_script_module *m);
/* 1 when this module's kernel may be launched as frame ranges at small voice counts (its frame body writes no memory: no
 * delay ring), 0 otherwise */
ZH_API int zh_script_module_ranges_ok(zh_script_module *m);
ZH_API int zh_script_module_get_state(zh_script_module *m, uint32_t *host_words);         /* [word][voice] */
ZH_API int zh_script_module_set_state(zh_script_module *m, const uint32_t *host_words);
ZH_API int zh_script_module_paint(zh_script_module *m, uint32_t span_start, uint32_t span_end, const zh_buf *outputs,
                                  zh_bool note_id_changed, const zh_script_param *params, uint32_t n_params, uint32_t flags);

/* The zangscript compiler itself (host side, no GPU work): src/zangscript/{tokenize,parse,codegen}.zig restated in
 * C++ (csrc/zscript_front.hip) with both backends (csrc/zscript_emit.hip): the reference's Zig text
 * (codegen_zig.zig; pins the front-end against the golden of src/zangscript/tests.zig:44-92) and the HIP source
 * zh_script_load takes.  `packages`: bit 0 = zang_builtin_package, bit 1 = modules_builtin_package
 * (builtins.zig:153-185).  A compile error returns ZH_ERR_INVALID with the message fail() would print
 * (`file:line:col: message`, source line, carets) in `err`.  Texts are malloc'ed: zh_zscript_free_text. */
typedef struct zh_zscript zh_zscript;
ZH_API int zh_zscript_compile(const char *text, const char *filename, uint32_t packages, zh_zscript **out, char *err, size_t err_cap);
ZH_API int zh_zscript_destroy(zh_zscript *z);
ZH_API void zh_zscript_free_text(char *text);
ZH_API int zh_zscript_generate_zig(zh_zscript *z, char **text_out);
/* only_csv: comma-separated exported module names, NULL = all; unroll: frames per unrolled chunk, 0 = automatic */
ZH_API int zh_zscript_generate_hip(zh_zscript *z, const char *only_csv, int unroll, char **text_out);
/* The same with more kernel forms per module.  ZH_ZSCRIPT_FORM_ROLES: next to zs_paint_<name> (one wave per 64 voices, the
 * whole frame body in every lane) a zs_paint_pc_<name> for FEW voices -- one workgroup of several waves per 64 voices, the
 * body's builtin calls (codegen.zig:70-121: the instruction list being scheduled) dealt to producer / recurrence / writer
 * waves that hand values on through LDS tiles, as the hand-written composites do (examples/modules.zig:130-187 is the
 * acceptance recipe).  Same operations on the same values in the same order: same bits.  zh_script_module_paint picks
 * the form (dispatch table rows script_pc / script_pc_maxv). */
/* ZH_ZSCRIPT_FORM_ROLES: for every module that has more than one role.  ZH_ZSCRIPT_FORM_ROLES_WORTH: only where the emitter's
 * estimate says it pays (a chain that can not be cut into frame ranges, the longest role well below the whole body): about half
 * the hiprtc time of the former (profiles/r06/script_compile_times.txt). */
#define ZH_ZSCRIPT_FORM_ROLES 1
#define ZH_ZSCRIPT_FORM_ROLES_WORTH 2
ZH_API int zh_zscript_generate_hip_forms(zh_zscript *z, const char *only_csv, int unroll, uint32_t forms, char **text_out);
/* per module of the last zh_zscript_generate_hip: what zh_script_module_create / _paint need */
ZH_API uint32_t zh_zscript_module_count(zh_zscript *z);
ZH_API int zh_zscript_module_info(zh_zscript *z, uint32_t i, char *name, size_t name_cap, uint32_t *state_words, uint32_t *noise_fields,
                                  uint32_t *n_params, char *error /* non-empty: the HIP backend cannot express this module */, size_t error_cap);
/* `pub const num_temps` of the generated Zig struct (codegen_zig.zig:518): what a reference host would allocate for this
 * module; the fused kernel itself needs none. */
ZH_API int zh_zscript_module_num_temps(zh_zscript *z, uint32_t i, uint32_t *num_temps);
ZH_API int zh_zscript_module_param(zh_zscript *z, uint32_t i, uint32_t p, char *name, size_t name_cap, char *kind, size_t kind_cap,
                                   char *enum_name, size_t enum_cap);

/* ---------------------------------------------------------------- event scheduling (host side; no GPU work)
 * The immediate caller of every paint: song / key events -> impulses ->
 * per-voice (span, params, note_id_changed) tuples.  A C++ restatement of src/zang/notes.zig and
 * src/zang/trigger.zig behind the same call shapes; `params` are opaque blobs of `params_size`
 * bytes (the Zig code is generic over NoteParamsType).  Slices returned by consume/dispatch stay
 * valid until the next call on the same object, like the Zig versions' internal arrays. */
typedef struct zh_impulse { uint64_t frame, note_id, event_id; } zh_impulse;                          /* notes.zig:58-62 */
typedef struct zh_iap { const zh_impulse *impulses; const void *paramses; uint64_t len; } zh_iap;     /* ImpulsesAndParamses :66-70 */
enum { ZH_MAX_IMPULSES = 32, ZH_MAX_PARAMS_SIZE = 64 };                                               /* notes.zig:73-74 */

typedef struct zh_impulse_queue zh_impulse_queue;                                                     /* notes.zig:72-128 */
ZH_API int zh_impulse_queue_create(uint32_t params_size, zh_impulse_queue **out);
ZH_API int zh_impulse_queue_destroy(zh_impulse_queue *q);
ZH_API int zh_impulse_queue_push(zh_impulse_queue *q, uint64_t impulse_frame, uint64_t note_id, const void *params);
ZH_API int zh_impulse_queue_consume(zh_impulse_queue *q, zh_iap *out);

typedef struct zh_note_tracker zh_note_tracker;                                                       /* notes.zig:138-207 */
/* song: n events, each {params blob, t (seconds, f32), note_id}; given as three parallel arrays (copied) */
ZH_API int zh_note_tracker_create(uint32_t params_size, uint64_t n_events, const void *paramses, const float *t,
                                  const uint64_t *note_ids, zh_note_tracker **out);
ZH_API int zh_note_tracker_destroy(zh_note_tracker *nt);
ZH_API int zh_note_tracker_reset(zh_note_tracker *nt);
ZH_API int zh_note_tracker_consume(zh_note_tracker *nt, float sample_rate, uint64_t span_start, uint64_t span_end, zh_iap *out);

typedef struct zh_polyphony_dispatcher zh_polyphony_dispatcher;                                       /* notes.zig:209-349 */
/* note_on is read from each params blob at byte `note_on_offset` (1 byte, non-zero = on) */
ZH_API int zh_polyphony_dispatcher_create(uint32_t polyphony, uint32_t params_size, uint32_t note_on_offset,
                                          zh_polyphony_dispatcher **out);
ZH_API int zh_polyphony_dispatcher_destroy(zh_polyphony_dispatcher *pd);
ZH_API int zh_polyphony_dispatcher_reset(zh_polyphony_dispatcher *pd);
ZH_API int zh_polyphony_dispatcher_dispatch(zh_polyphony_dispatcher *pd, zh_iap iap, zh_iap *out /*[polyphony]*/);

typedef struct zh_trigger zh_trigger;                                                                 /* trigger.zig:26-198 */
typedef struct zh_paint_span {                                                                        /* NewPaintReturnValue :50-54 */
    uint64_t start, end; uint32_t note_id_changed; uint32_t reserved; uint8_t params[64];
} zh_paint_span;
ZH_API int zh_trigger_create(uint32_t params_size, zh_trigger **out);
ZH_API int zh_trigger_destroy(zh_trigger *t);
ZH_API int zh_trigger_reset(zh_trigger *t);                                                           /* :62-64 */
ZH_API int zh_trigger_counter(zh_trigger *t, uint64_t span_start, uint64_t span_end, zh_iap iap);     /* :66-78 */
ZH_API int zh_trigger_next(zh_trigger *t, zh_paint_span *out);   /* :80-105; returns 1 = span produced, 0 = null, <0 error */

/* Voice(T)'s scheduling half (examples/example_song.zig:287-350): NoteTracker -> PolyphonyDispatcher(polyphony) ->
 * one Trigger per sub-voice.  zh_poly_voice_schedule makes the scheduling calls of `n_buffers` consecutive
 * Voice(T).paint invocations (buffer b has frames[b] frames; its sub-spans are shifted by the frames before it,
 * Trigger's carry-over splitting notes at every buffer edge exactly as per-buffer calls do) and records per
 * sub-voice the (span, params, note_id_changed) of every module.paint the reference would issue, laid out like
 * zh_span_table: [span][sub_voice], `max_spans` rows, counts[sub_voice].  ZH_ERR_INVALID if a list overflows. */
typedef struct zh_poly_voice zh_poly_voice;
ZH_API int zh_poly_voice_create(uint32_t polyphony, uint32_t params_size, uint32_t note_on_offset, uint64_t n_events,
                                const void *paramses, const float *t, const uint64_t *note_ids, zh_poly_voice **out);
ZH_API int zh_poly_voice_destroy(zh_poly_voice *pv);
ZH_API int zh_poly_voice_reset(zh_poly_voice *pv);
ZH_API int zh_poly_voice_schedule(zh_poly_voice *pv, float sample_rate, const uint32_t *frames, uint32_t n_buffers,
                                  uint32_t max_spans, uint32_t *counts, uint32_t *start, uint32_t *end, void *params,
                                  uint8_t *note_id_changed);

/* ---------------------------------------------------------------- single-voice host-pointer wrappers
 * The literal one-voice form of a Zig module call: `state` is the Zig struct (in/out), `outputs[0]` and every
 * input buffer are HOST float[>= span_end] slices exactly like zang's []f32, params are plain values.  Each call
 * stages through the device (upload, one-voice paint, download) and is synchronous: a drop-in for porting and
 * for checking a Zig caller against the GPU path, not a fast path (use the batched entry points for speed). */
typedef struct zh_hcob { uint32_t tag; float constant; const float *buffer; } zh_hcob;   /* host ConstantOrBuffer */
typedef struct zh_hcurve { uint32_t tag; float duration; } zh_hcurve;                     /* host PaintCurve */

typedef struct zh_sineosc_host_params { float sample_rate; zh_hcob freq, phase; } zh_sineosc_host_params;
typedef struct zh_pulseosc_host_params { float sample_rate; zh_hcob freq; float color; } zh_pulseosc_host_params;
typedef zh_pulseosc_host_params zh_trisawosc_host_params;
typedef struct zh_noise_host_params { uint32_t color; } zh_noise_host_params;
typedef struct zh_envelope_host_params { float sample_rate; zh_hcurve attack, decay, release; float sustain_volume; uint32_t note_on; } zh_envelope_host_params;
typedef struct zh_gate_host_params { uint32_t note_on; } zh_gate_host_params;
typedef struct zh_filter_host_params { const float *input; uint32_t type; zh_hcob cutoff, res; } zh_filter_host_params;
typedef struct zh_sampler_host_params {
    float sample_rate; uint64_t num_channels, sample_rate_in; uint32_t format; const uint8_t *data; uint64_t data_len;
    uint64_t channel; uint32_t loop;
} zh_sampler_host_params;
typedef struct zh_decimator_host_params { float sample_rate; const float *input; float fake_sample_rate; } zh_decimator_host_params;
typedef struct zh_distortion_host_params { const float *input; uint32_t type; float ingain, outgain, offset; } zh_distortion_host_params;

/* init() of the two modules whose Zig init() is not all-zeros (host side, no GPU work) */
ZH_API int zh_noise_state_init(zh_noise_state *state, uint64_t seed);          /* Noise.zig:25-32 with an explicit seed */
ZH_API int zh_decimator_state_init(zh_decimator_state *state);                  /* Decimator.zig:14-19 */

ZH_API int zh_sineosc_paint_host(zh_ctx *ctx, zh_sineosc_state *state, uint32_t span_start, uint32_t span_end, float *const *outputs, float *const *temps, uint32_t note_id_changed, const zh_sineosc_host_params *params);
ZH_API int zh_pulseosc_paint_host(zh_ctx *ctx, zh_pulseosc_state *state, uint32_t span_start, uint32_t span_end, float *const *outputs, float *const *temps, uint32_t note_id_changed, const zh_pulseosc_host_params *params);
ZH_API int zh_trisawosc_paint_host(zh_ctx *ctx, zh_trisawosc_state *state, uint32_t span_start, uint32_t span_end, float *const *outputs, float *const *temps, uint32_t note_id_changed, const zh_trisawosc_host_params *params);
ZH_API int zh_noise_paint_host(zh_ctx *ctx, zh_noise_state *state, uint32_t span_start, uint32_t span_end, float *const *outputs, float *const *temps, uint32_t note_id_changed, const zh_noise_host_params *params);
ZH_API int zh_envelope_paint_host(zh_ctx *ctx, zh_envelope_state *state, uint32_t span_start, uint32_t span_end, float *const *outputs, float *const *temps, uint32_t note_id_changed, const zh_envelope_host_params *params);
ZH_API int zh_gate_paint_host(zh_ctx *ctx, void *state_unused, uint32_t span_start, uint32_t span_end, float *const *outputs, float *const *temps, uint32_t note_id_changed, const zh_gate_host_params *params);
ZH_API int zh_filter_paint_host(zh_ctx *ctx, zh_filter_state *state, uint32_t span_start, uint32_t span_end, float *const *outputs, float *const *temps, uint32_t note_id_changed, const zh_filter_host_params *params);
ZH_API int zh_sampler_paint_host(zh_ctx *ctx, zh_sampler_state *state, uint32_t span_start, uint32_t span_end, float *const *outputs, float *const *temps, uint32_t note_id_changed, const zh_sampler_host_params *params);
ZH_API int zh_decimator_paint_host(zh_ctx *ctx, zh_decimator_state *state, uint32_t span_start, uint32_t span_end, float *const *outputs, float *const *temps, uint32_t note_id_changed, const zh_decimator_host_params *params);
ZH_API int zh_distortion_paint_host(zh_ctx *ctx, void *state_unused, uint32_t span_start, uint32_t span_end, float *const *outputs, float *const *temps, uint32_t note_id_changed, const zh_distortion_host_params *params);

#ifdef __cplusplus
}
#endif
#endif /* ZANG_HIP_H */
