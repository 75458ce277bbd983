// zang_hip.zig -- reference-side binding of libzang_hip.so (include/zang_hip.h).
//
// This is the file a zang maintainer adds to the Zig tree (e.g. as src/zang_hip.zig, linked with
// `exe.linkSystemLibrary("zang_hip")`).  It is NOT compiled in this repository: the build image has
// no Zig toolchain (see DESIGN.md 1).  The struct layouts below are the C layouts of the header;
// tests/test_abi.py checks the same layouts against the C compiler for the ctypes mirror.
//
// Each wrapper keeps zang's module interface (src/modules/SineOsc.zig:8-31)
//     paint(self, span, outputs, temps, note_id_changed, params)
// but `self` is a BATCH of n voices living on the GPU and outputs/temps are device images
// laid out [frame][voice] (zh_buf) instead of host []f32 slices.

const std = @import("std");

pub const Ctx = opaque {};
pub const Buf = extern struct { ptr: ?[*]f32, voices: u32, frames: u32, stride: u32, reserved: u32 = 0 };
pub const F32 = extern struct { value: f32 = 0, reserved: u32 = 0, per_voice: ?[*]const f32 = null };
pub const Bool = extern struct { value: u32 = 0, reserved: u32 = 0, per_voice: ?[*]const u8 = null };
pub const Cob = extern struct { tag: u32, reserved: u32 = 0, constant: F32 = .{}, buffer: Buf = std.mem.zeroes(Buf) };
pub const Curve = extern struct { tag: u32, reserved: u32 = 0, duration: F32 = .{} };

pub const PAINT_ADD: u32 = 0;
pub const PAINT_ZERO_FIRST: u32 = 1;
pub const PAINT_PARAMS_UNCHANGED: u32 = 4; // params are what this module's previous paint got (include/zang_hip.h)

pub extern fn zh_create(out: *?*Ctx, device: c_int) c_int;
pub extern fn zh_destroy(ctx: *Ctx) c_int;
pub extern fn zh_sync(ctx: *Ctx) c_int;
pub extern fn zh_buf_alloc(ctx: *Ctx, out: *Buf, voices: u32, frames: u32) c_int;
pub extern fn zh_buf_free(ctx: *Ctx, buf: *Buf) c_int;
pub extern fn zh_buf_upload_voice(ctx: *Ctx, dst: Buf, voice: u32, host: [*]const f32, frames: u32) c_int;
pub extern fn zh_buf_download_voice(ctx: *Ctx, host: [*]f32, src: Buf, voice: u32, frames: u32) c_int;

// basics.zig:12-78
pub extern fn zh_zero(ctx: *Ctx, start: u32, end: u32, dest: Buf) c_int;
pub extern fn zh_set(ctx: *Ctx, start: u32, end: u32, dest: Buf, a: F32) c_int;
pub extern fn zh_copy(ctx: *Ctx, start: u32, end: u32, dest: Buf, src: Buf) c_int;
pub extern fn zh_add(ctx: *Ctx, start: u32, end: u32, dest: Buf, a: Buf, b: Buf) c_int;
pub extern fn zh_add_into(ctx: *Ctx, start: u32, end: u32, dest: Buf, src: Buf) c_int;
pub extern fn zh_add_scalar(ctx: *Ctx, start: u32, end: u32, dest: Buf, a: Buf, b: F32) c_int;
pub extern fn zh_add_scalar_into(ctx: *Ctx, start: u32, end: u32, dest: Buf, a: F32) c_int;
pub extern fn zh_multiply(ctx: *Ctx, start: u32, end: u32, dest: Buf, a: Buf, b: Buf) c_int;
pub extern fn zh_multiply_with(ctx: *Ctx, start: u32, end: u32, dest: Buf, a: Buf) c_int;
pub extern fn zh_multiply_scalar(ctx: *Ctx, start: u32, end: u32, dest: Buf, a: Buf, b: F32) c_int;
pub extern fn zh_multiply_with_scalar(ctx: *Ctx, start: u32, end: u32, dest: Buf, a: F32) c_int;
pub extern fn zh_mixdown_voices(ctx: *Ctx, start: u32, end: u32, dst: [*]f32, src: Buf, flags: u32) c_int;

// multi-GPU mixdown exchange without a collective library: the root process owns one slot per rank in its GPU's memory,
// the other processes map it (64-byte handle over any host channel) and point their mixdown at `base + rank * slot_bytes`
pub extern fn zh_ipc_alloc(ctx: *Ctx, bytes: usize, dev_ptr: *?*anyopaque, handle64: *[64]u8) c_int;
pub extern fn zh_ipc_open(ctx: *Ctx, handle64: *const [64]u8, dev_ptr: *?*anyopaque) c_int;
pub extern fn zh_ipc_close(ctx: *Ctx, dev_ptr: ?*anyopaque) c_int;
pub extern fn zh_sum_slots(ctx: *Ctx, dst: [*]f32, slots: [*]const f32, n_slots: u32, slot_stride_floats: usize, n: usize, flags: u32) c_int;

// zangscript modules (include/zang_hip.h "zangscript modules"): HIP source from `python -m zang_amd.zangc`
pub const Script = opaque {};
pub const ScriptModule = opaque {};
pub const SP_CONSTANT: u32 = 0;
pub const SP_BOOLEAN: u32 = 1;
pub const SP_COB: u32 = 2;
pub const SP_BUFFER: u32 = 3;
pub const SP_ENUM: u32 = 4;
pub const SP_CURVE: u32 = 5;
pub const ScriptParam = extern struct {
    kind: u32,
    u: u32 = 0,
    f: f32 = 0,
    is_buffer: u32 = 0,
    pf: ?[*]const f32 = null,
    pb: ?[*]const u8 = null,
    stride: u32 = 0,
    reserved: u32 = 0,
};
pub extern fn zh_script_load(ctx: *Ctx, hip_source: [*:0]const u8, out: *?*Script, log: ?[*]u8, log_cap: usize) c_int;
pub extern fn zh_script_destroy(s: *Script) c_int;
pub extern fn zh_script_module_create(s: *Script, name: [*:0]const u8, n_voices: u32, state_words: u32, first_seed: u64, out: *?*ScriptModule) c_int;
pub extern fn zh_script_module_destroy(m: *ScriptModule) c_int;
pub extern fn zh_script_module_paint(m: *ScriptModule, start: u32, end: u32, outputs: [*]const Buf, note_id_changed: Bool, params: [*]const ScriptParam, n_params: u32, flags: u32) c_int;

// PulseOsc (src/modules/PulseOsc.zig)
pub const PulseOscHandle = opaque {};
pub const PulseOscParams = extern struct { sample_rate: f32, reserved: u32 = 0, freq: Cob, color: F32 };
pub extern fn zh_pulseosc_create(ctx: *Ctx, n_voices: u32, out: *?*PulseOscHandle) c_int;
pub extern fn zh_pulseosc_destroy(m: *PulseOscHandle) c_int;
pub extern fn zh_pulseosc_paint(m: *PulseOscHandle, start: u32, end: u32, outputs: [*]const Buf, temps: ?[*]const Buf, note_id_changed: Bool, params: *const PulseOscParams, flags: u32) c_int;
// n_buffers consecutive paint calls with the same span and params (the host's loop over 1024-frame buffers,
// examples/write_wav.zig:58-66), buffer b into outputs[b]: one launch when the frequency is constant
pub extern fn zh_pulseosc_paint_batch(m: *PulseOscHandle, start: u32, end: u32, outputs: [*]const Buf, n_buffers: u32, params: *const PulseOscParams, flags: u32) c_int;

// Filter (src/modules/Filter.zig)
pub const FilterHandle = opaque {};
pub const FilterParams = extern struct { input: Buf, type: u32, reserved: u32 = 0, cutoff: Cob, res: Cob };
pub extern fn zh_filter_create(ctx: *Ctx, n_voices: u32, out: *?*FilterHandle) c_int;
pub extern fn zh_filter_destroy(m: *FilterHandle) c_int;
pub extern fn zh_filter_paint(m: *FilterHandle, start: u32, end: u32, outputs: [*]const Buf, temps: ?[*]const Buf, note_id_changed: Bool, params: *const FilterParams, flags: u32) c_int;
pub extern fn zh_filter_cutoff_from_frequency(ctx: *Ctx, n: u32, cutoff_out_dev: [*]f32, frequency_dev: [*]const f32, sample_rate: f32) c_int;
// std.math.sin / cos / pow (f32) elementwise on device arrays, same bits as the modules
pub extern fn zh_sin(ctx: *Ctx, n: u32, out: [*]f32, x: [*]const f32) c_int;
pub extern fn zh_cos(ctx: *Ctx, n: u32, out: [*]f32, x: [*]const f32) c_int;
pub extern fn zh_atan(ctx: *Ctx, n: u32, out: [*]f32, x: [*]const f32) c_int;
pub extern fn zh_pow(ctx: *Ctx, n: u32, out: [*]f32, x: [*]const f32, y: [*]const f32) c_int;

// Envelope (src/modules/Envelope.zig)
pub const EnvelopeHandle = opaque {};
pub const EnvelopeParams = extern struct { sample_rate: f32, reserved: u32 = 0, attack: Curve, decay: Curve, release: Curve, sustain_volume: F32, note_on: Bool };
pub extern fn zh_envelope_create(ctx: *Ctx, n_voices: u32, out: *?*EnvelopeHandle) c_int;
pub extern fn zh_envelope_destroy(m: *EnvelopeHandle) c_int;
pub extern fn zh_envelope_paint(m: *EnvelopeHandle, start: u32, end: u32, outputs: [*]const Buf, temps: ?[*]const Buf, note_id_changed: Bool, params: *const EnvelopeParams, flags: u32) c_int;

// ... SineOsc, TriSawOsc, Noise, Gate, Sampler, Decimator, Distortion, NiceInstrument and
// PMOscInstrument follow the same pattern: see include/zang_hip.h for their params structs.

pub fn constant(x: f32) Cob {
    return .{ .tag = 0, .constant = .{ .value = x } };
}
pub fn constantPerVoice(xs: [*]const f32) Cob {
    return .{ .tag = 0, .constant = .{ .per_voice = xs } };
}
pub fn buffer(b: Buf) Cob {
    return .{ .tag = 1, .buffer = b };
}

fn check(rc: c_int) void {
    // paint() cannot fail in zang (returns void); a HIP error here is a programming error
    if (rc != 0) std.debug.panic("zang_hip: error {d}", .{rc});
}

/// Drop-in shaped like `mod.PulseOsc` for a batch of `n` voices on the GPU.
pub const PulseOsc = struct {
    pub const num_outputs = 1;
    pub const num_temps = 0;
    pub const Params = struct { sample_rate: f32, freq: Cob, color: F32 };

    handle: *PulseOscHandle,

    pub fn init(ctx: *Ctx, n_voices: u32) PulseOsc {
        var h: ?*PulseOscHandle = null;
        check(zh_pulseosc_create(ctx, n_voices, &h));
        return .{ .handle = h.? };
    }

    pub fn paint(
        self: *PulseOsc,
        span: struct { start: usize, end: usize }, // zang.Span
        outputs: [num_outputs]Buf,
        temps: [num_temps]Buf,
        note_id_changed: bool,
        params: Params,
    ) void {
        _ = temps;
        const p = PulseOscParams{ .sample_rate = params.sample_rate, .freq = params.freq, .color = params.color };
        check(zh_pulseosc_paint(self.handle, @intCast(span.start), @intCast(span.end), &outputs, null, .{ .value = @intFromBool(note_id_changed) }, &p, PAINT_ADD));
    }
};
