// dump_vectors.zig -- runs the REFERENCE's own paint() code (dbandstra/zang, Zig 0.12 / 0.13) on fixed inputs and
// writes inputs + outputs + final state as vector files, one per case, for tests/test_zig_vectors.py to hold the
// repo's oracle (oracle/zang_oracle.c) against.  This is the one-command route from "parity unpinned" to pinned.
//
// IT HAS NEVER BEEN COMPILED: the image this repo is built in has no Zig toolchain (SURVEY.md 8c).  It is written
// against Zig 0.12 / 0.13 syntax and the reference's public declarations:
//   zang.Span.init, zang.zero / multiply / mixDown, zang.constant / zang.buffer, zang.PaintCurve      (src/zang.zig)
//   mod.{SineOsc,PulseOsc,TriSawOsc,Noise,Envelope,Gate,Filter,Sampler,Decimator,Distortion}          (src/modules.zig)
//   NiceInstrument, PMOscInstrument                                                                    (examples/modules.zig)
// A maintainer with Zig fixes whatever a newer compiler rejects; the file format and the case list are the contract.
//
// Build and run on a machine that has Zig and a checkout of the reference at $REF (see README.md beside this file):
//   zig build-exe -O ReleaseSafe \
//     --dep zang --dep modules --dep exmods -Mroot=tools/zig_oracle/dump_vectors.zig \
//     -Mzang=$REF/src/zang.zig --dep zang -Mmodules=$REF/src/modules.zig -Mzang-12tet=$REF/src/zang-12tet.zig \
//     --dep zang --dep modules --dep zang-12tet -Mexmods=$REF/examples/modules.zig
//   ./dump_vectors tests/golden/zig
// Only the vector files (data) are committed; no reference source travels with them.
//
// File format (little endian): "ZGV1", then records until EOF:
//   u32 name_len, name bytes, u32 type (0 f32, 1 u32, 2 u64, 3 u8), u32 count, count * sizeof(type) bytes.
// What each case computes is fixed by its NAME; tests/test_zig_vectors.py holds the same table.
const std = @import("std");
const zang = @import("zang");
const mod = @import("modules");
const ex = @import("exmods");

const F = 1024; // frames per buffer (examples: AUDIO_BUFFER_SIZE)
const SR: f32 = 48000.0;

const Sub = struct { s: usize, e: usize, on: bool, nic: bool };
const spans3 = [_][2]usize{ .{ 0, 200 }, .{ 200, 777 }, .{ 777, 1024 } }; // SURVEY.md appendix B
// two buffers: note on with a new id, held, released inside the first buffer; silence in the second
const note_script = [_]Sub{
    .{ .s = 0, .e = 200, .on = true, .nic = true },
    .{ .s = 200, .e = 777, .on = true, .nic = false },
    .{ .s = 777, .e = 1024, .on = false, .nic = false },
    .{ .s = 1024, .e = 2048, .on = false, .nic = false },
};
// retrigger: a second note id while the first is still sounding
const retrigger_script = [_]Sub{
    .{ .s = 0, .e = 300, .on = true, .nic = true },
    .{ .s = 300, .e = 600, .on = true, .nic = true },
    .{ .s = 600, .e = 1024, .on = false, .nic = false },
    .{ .s = 1024, .e = 2048, .on = true, .nic = true },
};

// ---------------------------------------------------------------- inputs: SplitMix64 -> uniform f32
fn splitmix(state: *u64) u64 {
    state.* +%= 0x9e3779b97f4a7c15;
    var z = state.*;
    z = (z ^ (z >> 30)) *% 0xbf58476d1ce4e5b9;
    z = (z ^ (z >> 27)) *% 0x94d049bb133111eb;
    return z ^ (z >> 31);
}
fn fill(buf: []f32, seed: u64, lo: f32, hi: f32) void {
    var st: u64 = seed;
    for (buf) |*x| {
        const u: f32 = @as(f32, @floatFromInt(splitmix(&st) >> 40)) / 16777216.0; // 24 bits: exact in f32
        x.* = lo + (hi - lo) * u;
    }
}

// ---------------------------------------------------------------- output
const Out = struct {
    file: std.fs.File,

    fn open(dir: std.fs.Dir, name: []const u8) !Out {
        var buf: [128]u8 = undefined;
        const path = try std.fmt.bufPrint(&buf, "{s}.zgv", .{name});
        const f = try dir.createFile(path, .{});
        try f.writeAll("ZGV1");
        return .{ .file = f };
    }
    fn put(self: *Out, name: []const u8, comptime T: type, data: []const T) !void {
        const w = self.file.writer();
        try w.writeInt(u32, @intCast(name.len), .little);
        try w.writeAll(name);
        const code: u32 = switch (T) {
            f32 => 0,
            u32 => 1,
            u64 => 2,
            u8 => 3,
            else => @compileError("record type"),
        };
        try w.writeInt(u32, code, .little);
        try w.writeInt(u32, @intCast(data.len), .little);
        try w.writeAll(std.mem.sliceAsBytes(data));
    }
    fn f32s(self: *Out, name: []const u8, data: []const f32) !void {
        try self.put(name, f32, data);
    }
    fn close(self: *Out) void {
        self.file.close();
    }
};

fn cob(is_buffer: bool, c: f32, buf: []const f32) zang.ConstantOrBuffer {
    return if (is_buffer) zang.buffer(buf) else zang.constant(c);
}

// ---------------------------------------------------------------- cases
fn sineosc(dir: std.fs.Dir) !void {
    const names = [_][]const u8{ "sineosc_cc", "sineosc_cb", "sineosc_bc", "sineosc_bb" };
    for (names, 0..) |name, k| {
        const fb = (k & 2) != 0;
        const pb = (k & 1) != 0;
        var out: [F]f32 = undefined;
        var freq: [F]f32 = undefined;
        var phase: [F]f32 = undefined;
        fill(&out, 100 + k, -1.0, 1.0);
        fill(&freq, 200 + k, 20.0, 2000.0);
        fill(&phase, 300 + k, -1.0, 1.0);
        var o = try Out.open(dir, name);
        defer o.close();
        try o.f32s("out0", &out);
        try o.f32s("freq", &freq);
        try o.f32s("phase", &phase);
        var m = mod.SineOsc.init();
        for (spans3) |sp| {
            m.paint(zang.Span.init(sp[0], sp[1]), .{&out}, .{}, false, .{
                .sample_rate = SR,
                .freq = cob(fb, 440.0, &freq),
                .phase = cob(pb, 0.25, &phase),
            });
        }
        try o.f32s("out", &out);
        try o.f32s("state_t", &[_]f32{m.t});
    }
}

fn oscillators(dir: std.fs.Dir) !void {
    // constant frequency: (name, freq, color)
    const Const = struct { name: []const u8, freq: f32, color: f32 };
    const consts = [_]Const{
        .{ .name = "const_c0", .freq = 440.0, .color = 0.0 },
        .{ .name = "const_c03", .freq = 440.0, .color = 0.3 },
        .{ .name = "const_c05", .freq = 1234.5, .color = 0.5 },
        .{ .name = "const_c09", .freq = 97.0, .color = 0.9 },
        .{ .name = "const_c1", .freq = 5999.0, .color = 1.0 },
        .{ .name = "const_silent_hi", .freq = 6000.5, .color = 0.5 }, // > sr/8: paints nothing
        .{ .name = "const_silent_neg", .freq = -1.0, .color = 0.5 },
    };
    for (consts, 0..) |c, k| {
        inline for (.{ "pulseosc_", "trisawosc_" }, 0..) |prefix, which| {
            var nb: [64]u8 = undefined;
            const name = try std.fmt.bufPrint(&nb, "{s}{s}", .{ prefix, c.name });
            var out: [F]f32 = undefined;
            fill(&out, 400 + k, -1.0, 1.0);
            var o = try Out.open(dir, name);
            defer o.close();
            try o.f32s("out0", &out);
            try o.f32s("params", &[_]f32{ SR, c.freq, c.color });
            if (which == 0) {
                var m = mod.PulseOsc.init();
                for (spans3) |sp| m.paint(zang.Span.init(sp[0], sp[1]), .{&out}, .{}, false, .{ .sample_rate = SR, .freq = zang.constant(c.freq), .color = c.color });
                try o.put("state_cnt", u32, &[_]u32{m.cnt});
            } else {
                var m = mod.TriSawOsc.init();
                for (spans3) |sp| m.paint(zang.Span.init(sp[0], sp[1]), .{&out}, .{}, false, .{ .sample_rate = SR, .freq = zang.constant(c.freq), .color = c.color });
                try o.put("state_cnt", u32, &[_]u32{m.cnt});
                try o.f32s("state_t", &[_]f32{m.t});
            }
            try o.f32s("out", &out);
        }
    }
    // controlled frequency (a buffer, with out-of-range samples): colors below / between / above the saw thresholds
    const colors = [_]f32{ 0.1, 0.5, 0.9 };
    for (colors, 0..) |color, k| {
        inline for (.{ "pulseosc_buf", "trisawosc_buf" }, 0..) |prefix, which| {
            var nb: [64]u8 = undefined;
            const name = try std.fmt.bufPrint(&nb, "{s}_{d}", .{ prefix, k });
            var out: [F]f32 = undefined;
            var freq: [F]f32 = undefined;
            fill(&out, 500 + k, -1.0, 1.0);
            fill(&freq, 600 + k, -200.0, 7000.0);
            var o = try Out.open(dir, name);
            defer o.close();
            try o.f32s("out0", &out);
            try o.f32s("freq", &freq);
            try o.f32s("params", &[_]f32{ SR, color });
            if (which == 0) {
                var m = mod.PulseOsc.init();
                for (spans3) |sp| m.paint(zang.Span.init(sp[0], sp[1]), .{&out}, .{}, false, .{ .sample_rate = SR, .freq = zang.buffer(&freq), .color = color });
                try o.put("state_cnt", u32, &[_]u32{m.cnt});
            } else {
                var m = mod.TriSawOsc.init();
                for (spans3) |sp| m.paint(zang.Span.init(sp[0], sp[1]), .{&out}, .{}, false, .{ .sample_rate = SR, .freq = zang.buffer(&freq), .color = color });
                try o.put("state_cnt", u32, &[_]u32{m.cnt});
                try o.f32s("state_t", &[_]f32{m.t});
            }
            try o.f32s("out", &out);
        }
    }
}

// Noise.init() takes its seed from a process-wide counter (Noise.zig:9,26): these are the FIRST four Noise instances of
// this process, seeds 0..3, and nothing else in this program may call Noise.init().
fn noise(dir: std.fs.Dir) !void {
    const names = [_][]const u8{ "noise_white_seed0", "noise_pink_seed1", "noise_white_seed2", "noise_pink_seed3" };
    for (names, 0..) |name, k| {
        var m = mod.Noise.init();
        var o = try Out.open(dir, name);
        defer o.close();
        var out: [2 * F]f32 = undefined;
        fill(&out, 700 + k, -1.0, 1.0);
        try o.f32s("out0", &out);
        // two consecutive paints (the second shows pink's taps restarting, Noise.zig:68), three sub-spans in the first
        for (spans3) |sp| {
            if ((k & 1) == 0) m.paint(zang.Span.init(sp[0], sp[1]), .{out[0..F]}, .{}, false, .{ .color = .white }) else m.paint(zang.Span.init(sp[0], sp[1]), .{out[0..F]}, .{}, false, .{ .color = .pink });
        }
        if ((k & 1) == 0) m.paint(zang.Span.init(0, F), .{out[F..]}, .{}, false, .{ .color = .white }) else m.paint(zang.Span.init(0, F), .{out[F..]}, .{}, false, .{ .color = .pink });
        try o.f32s("out", &out);
        try o.put("state_s", u64, &m.r.s); // Xoshiro256.s: [4]u64
    }
}

fn curveOf(kind: usize, dur: f32) zang.PaintCurve {
    return switch (kind) {
        0 => .instantaneous,
        1 => .{ .linear = dur },
        2 => .{ .squared = dur },
        else => .{ .cubed = dur },
    };
}

fn envelope(dir: std.fs.Dir) !void {
    // (attack kind, decay kind, release kind) x sustain; durations chosen so stages end inside spans
    const combos = [_][3]usize{ .{ 3, 3, 3 }, .{ 1, 2, 3 }, .{ 0, 1, 1 }, .{ 2, 0, 2 }, .{ 1, 1, 0 }, .{ 0, 0, 0 }, .{ 3, 1, 2 } };
    const sustains = [_]f32{ 0.6, 1.0 };
    const scripts = [_][]const Sub{ &note_script, &retrigger_script };
    for (combos, 0..) |cmb, ci| {
        for (sustains, 0..) |sus, si| {
            for (scripts, 0..) |script, ki| {
                var nb: [64]u8 = undefined;
                const name = try std.fmt.bufPrint(&nb, "envelope_{d}_{d}_{d}", .{ ci, si, ki });
                var out: [2 * F]f32 = undefined;
                fill(&out, 800 + ci * 10 + si * 2 + ki, -1.0, 1.0);
                var o = try Out.open(dir, name);
                defer o.close();
                try o.f32s("out0", &out);
                try o.f32s("params", &[_]f32{ SR, @floatFromInt(cmb[0]), 0.002, @floatFromInt(cmb[1]), 0.004, @floatFromInt(cmb[2]), 0.003, sus });
                var m = mod.Envelope.init();
                for (script) |st| {
                    const base: usize = if (st.s >= F) F else 0; // second buffer: its own 1024-frame slice, spans from 0
                    m.paint(zang.Span.init(st.s - base, st.e - base), .{out[base .. base + F]}, .{}, st.nic, .{
                        .sample_rate = SR,
                        .attack = curveOf(cmb[0], 0.002),
                        .decay = curveOf(cmb[1], 0.004),
                        .release = curveOf(cmb[2], 0.003),
                        .sustain_volume = sus,
                        .note_on = st.on,
                    });
                }
                try o.f32s("out", &out);
                try o.put("state_stage", u32, &[_]u32{@intFromEnum(m.state)});
                try o.f32s("state_painter", &[_]f32{ m.painter.t, m.painter.last_value, m.painter.start });
            }
        }
    }
}

fn gate(dir: std.fs.Dir) !void {
    var out: [F]f32 = undefined;
    fill(&out, 900, -1.0, 1.0);
    var o = try Out.open(dir, "gate");
    defer o.close();
    try o.f32s("out0", &out);
    var m = mod.Gate.init();
    m.paint(zang.Span.init(0, 200), .{&out}, .{}, false, .{ .note_on = true });
    m.paint(zang.Span.init(200, 777), .{&out}, .{}, false, .{ .note_on = false });
    m.paint(zang.Span.init(777, 1024), .{&out}, .{}, false, .{ .note_on = true });
    try o.f32s("out", &out);
}

fn filter(dir: std.fs.Dir) !void {
    var k: usize = 0;
    for (std.enums.values(mod.Filter.Type)) |t| {
        // every type with both params constant; low_pass and notch also with the three buffer combinations
        const paths: usize = if (t == .low_pass or t == .notch) 4 else 1;
        var p: usize = 0;
        while (p < paths) : (p += 1) {
            var nb: [64]u8 = undefined;
            const name = try std.fmt.bufPrint(&nb, "filter_{s}_{d}", .{ @tagName(t), p });
            var out: [F]f32 = undefined;
            var input: [F]f32 = undefined;
            var cutoff: [F]f32 = undefined;
            var res: [F]f32 = undefined;
            fill(&out, 1000 + k, -1.0, 1.0);
            fill(&input, 1100 + k, -1.0, 1.0);
            fill(&cutoff, 1200 + k, -0.1, 1.1); // beyond [0, 1]: the clamp (Filter.zig:114,126)
            fill(&res, 1300 + k, -0.1, 1.1);
            var o = try Out.open(dir, name);
            defer o.close();
            try o.f32s("out0", &out);
            try o.f32s("input", &input);
            try o.f32s("cutoff", &cutoff);
            try o.f32s("res", &res);
            try o.put("type", u32, &[_]u32{@intFromEnum(t)});
            var m = mod.Filter.init();
            for (spans3) |sp| {
                m.paint(zang.Span.init(sp[0], sp[1]), .{&out}, .{}, false, .{
                    .input = &input,
                    .type = t,
                    .cutoff = cob((p & 2) != 0, 0.3, &cutoff),
                    .res = cob((p & 1) != 0, 0.5, &res),
                });
            }
            try o.f32s("out", &out);
            try o.f32s("state_lb", &[_]f32{ m.l, m.b });
            k += 1;
        }
    }
    // cutoffFromFrequency (Filter.zig:20-23) over a frequency grid
    var freqs: [512]f32 = undefined;
    var cut: [512]f32 = undefined;
    fill(&freqs, 1400, 0.0, 26000.0);
    for (freqs, 0..) |f, i| cut[i] = mod.Filter.cutoffFromFrequency(f, SR);
    var o = try Out.open(dir, "filter_cutoff_from_frequency");
    defer o.close();
    try o.f32s("freq", &freqs);
    try o.f32s("out", &cut);
}

fn decimator(dir: std.fs.Dir) !void {
    const fakes = [_]f32{ 24000.0, 6000.0, 11025.0, 48000.0, 96000.0, 0.0, -5.0 };
    for (fakes, 0..) |fake, k| {
        var nb: [64]u8 = undefined;
        const name = try std.fmt.bufPrint(&nb, "decimator_{d}", .{k});
        var out: [F]f32 = undefined;
        var input: [F]f32 = undefined;
        fill(&out, 1500 + k, -1.0, 1.0);
        fill(&input, 1600 + k, -1.0, 1.0);
        var o = try Out.open(dir, name);
        defer o.close();
        try o.f32s("out0", &out);
        try o.f32s("input", &input);
        try o.f32s("params", &[_]f32{ SR, fake });
        var m = mod.Decimator.init();
        for (spans3) |sp| m.paint(zang.Span.init(sp[0], sp[1]), .{&out}, .{}, false, .{ .sample_rate = SR, .input = &input, .fake_sample_rate = fake });
        try o.f32s("out", &out);
        try o.f32s("state", &[_]f32{ m.dval, m.dcount });
    }
}

fn distortion(dir: std.fs.Dir) !void {
    const P = struct { ingain: f32, outgain: f32, offset: f32 };
    const ps = [_]P{ .{ .ingain = 0.5, .outgain = 0.7, .offset = 0.1 }, .{ .ingain = 0.25, .outgain = 1.0, .offset = 0.0 }, .{ .ingain = 0.9, .outgain = 0.3, .offset = -0.4 } };
    for (ps, 0..) |p, k| {
        for (std.enums.values(mod.Distortion.Type)) |t| {
            var nb: [64]u8 = undefined;
            const name = try std.fmt.bufPrint(&nb, "distortion_{s}_{d}", .{ @tagName(t), k });
            var out: [F]f32 = undefined;
            var input: [F]f32 = undefined;
            fill(&out, 1700 + k, -1.0, 1.0);
            fill(&input, 1800 + k, -1.5, 1.5);
            var o = try Out.open(dir, name);
            defer o.close();
            try o.f32s("out0", &out);
            try o.f32s("input", &input);
            try o.f32s("params", &[_]f32{ p.ingain, p.outgain, p.offset });
            try o.put("type", u32, &[_]u32{@intFromEnum(t)});
            var m = mod.Distortion.init();
            for (spans3) |sp| m.paint(zang.Span.init(sp[0], sp[1]), .{&out}, .{}, false, .{ .input = &input, .type = t, .ingain = p.ingain, .outgain = p.outgain, .offset = p.offset });
            try o.f32s("out", &out);
        }
    }
}

fn sampler(dir: std.fs.Dir) !void {
    // 2-channel signed 16-bit PCM, 300 frames; played at its own rate (integer copy path) and at 44100 -> 48000
    // (interpolating path), looped and not, both channels (Sampler.zig:77-136)
    var pcm: [300 * 2 * 2]u8 = undefined;
    var st: u64 = 1900;
    for (&pcm) |*b| b.* = @truncate(splitmix(&st));
    const rates = [_]usize{ 48000, 44100 };
    var k: usize = 0;
    for (rates) |rate| {
        for ([_]bool{ false, true }) |loop| {
            for ([_]usize{ 0, 1 }) |channel| {
                var nb: [64]u8 = undefined;
                const name = try std.fmt.bufPrint(&nb, "sampler_{d}", .{k});
                var out: [F]f32 = undefined;
                fill(&out, 2000 + k, -1.0, 1.0);
                var o = try Out.open(dir, name);
                defer o.close();
                try o.f32s("out0", &out);
                try o.put("pcm", u8, &pcm);
                try o.put("params", u32, &[_]u32{ 2, @intCast(rate), @intFromEnum(mod.Sampler.Format.signed16_lsb), @intCast(channel), @intFromBool(loop) });
                var m = mod.Sampler.init();
                for (spans3, 0..) |sp, si| {
                    m.paint(zang.Span.init(sp[0], sp[1]), .{&out}, .{}, si == 0, .{
                        .sample_rate = SR,
                        .sample = .{ .num_channels = 2, .sample_rate = rate, .format = .signed16_lsb, .data = &pcm },
                        .channel = channel,
                        .loop = loop,
                    });
                }
                try o.f32s("out", &out);
                try o.f32s("state_t", &[_]f32{m.t});
                k += 1;
            }
        }
    }
}

fn instruments(dir: std.fs.Dir) !void {
    const scripts = [_][]const Sub{ &note_script, &retrigger_script };
    const freqs = [_]f32{ 440.0, 55.0, 2793.83 };
    for (scripts, 0..) |script, ki| {
        for (freqs, 0..) |freq, fi| {
            {
                var nb: [64]u8 = undefined;
                const name = try std.fmt.bufPrint(&nb, "nice_{d}_{d}", .{ ki, fi });
                var out: [2 * F]f32 = undefined;
                var t0: [F]f32 = undefined;
                var t1: [F]f32 = undefined;
                fill(&out, 2100 + ki * 4 + fi, -1.0, 1.0);
                var o = try Out.open(dir, name);
                defer o.close();
                try o.f32s("out0", &out);
                const color: f32 = 0.3 + 0.2 * @as(f32, @floatFromInt(fi));
                try o.f32s("params", &[_]f32{ SR, freq, color });
                var m = ex.NiceInstrument.init(color);
                for (script) |st| {
                    const base: usize = if (st.s >= F) F else 0;
                    m.paint(zang.Span.init(st.s - base, st.e - base), .{out[base .. base + F]}, .{ &t0, &t1 }, st.nic, .{ .sample_rate = SR, .freq = freq, .note_on = st.on });
                }
                try o.f32s("out", &out);
                try o.put("state_cnt", u32, &[_]u32{m.osc.cnt});
                try o.f32s("state_lb", &[_]f32{ m.flt.l, m.flt.b });
                try o.put("state_stage", u32, &[_]u32{@intFromEnum(m.env.state)});
                try o.f32s("state_painter", &[_]f32{ m.env.painter.t, m.env.painter.last_value, m.env.painter.start });
            }
            {
                var nb: [64]u8 = undefined;
                const name = try std.fmt.bufPrint(&nb, "pmosc_{d}_{d}", .{ ki, fi });
                var out: [2 * F]f32 = undefined;
                var t0: [F]f32 = undefined;
                var t1: [F]f32 = undefined;
                var t2: [F]f32 = undefined;
                fill(&out, 2200 + ki * 4 + fi, -1.0, 1.0);
                var o = try Out.open(dir, name);
                defer o.close();
                try o.f32s("out0", &out);
                try o.f32s("params", &[_]f32{ SR, freq * 0.5, 0.4 }); // example_song.zig:28-42: freq * 0.5, release 0.4
                var m = ex.PMOscInstrument.init(0.4);
                for (script) |st| {
                    const base: usize = if (st.s >= F) F else 0;
                    m.paint(zang.Span.init(st.s - base, st.e - base), .{out[base .. base + F]}, .{ &t0, &t1, &t2 }, st.nic, .{ .sample_rate = SR, .freq = freq * 0.5, .note_on = st.on });
                }
                try o.f32s("out", &out);
                try o.f32s("state_t", &[_]f32{ m.osc.carrier.t, m.osc.modulator.t });
                try o.put("state_stage", u32, &[_]u32{@intFromEnum(m.env.state)});
                try o.f32s("state_painter", &[_]f32{ m.env.painter.t, m.env.painter.last_value, m.env.painter.start });
            }
            {
                // examples/modules.zig:130-187 (round 5: the recipes tests/test_gpu_script_composites.py holds the generated kernels to)
                var nb: [64]u8 = undefined;
                const name = try std.fmt.bufPrint(&nb, "fsaw_{d}_{d}", .{ ki, fi });
                var out: [2 * F]f32 = undefined;
                var t0: [F]f32 = undefined;
                var t1: [F]f32 = undefined;
                var t2: [F]f32 = undefined;
                fill(&out, 2600 + ki * 4 + fi, -1.0, 1.0);
                var o = try Out.open(dir, name);
                defer o.close();
                try o.f32s("out0", &out);
                try o.f32s("params", &[_]f32{ SR, freq, 0.0 });
                var m = ex.FilteredSawtoothInstrument.init();
                for (script) |st| {
                    const base: usize = if (st.s >= F) F else 0;
                    m.paint(zang.Span.init(st.s - base, st.e - base), .{out[base .. base + F]}, .{ &t0, &t1, &t2 }, st.nic, .{ .sample_rate = SR, .freq = zang.constant(freq), .note_on = st.on });
                }
                try o.f32s("out", &out);
                try o.put("state_cnt", u32, &[_]u32{m.osc.cnt});
                try o.f32s("state_lb", &[_]f32{ m.flt.l, m.flt.b });
                try o.put("state_stage", u32, &[_]u32{@intFromEnum(m.env.state)});
                try o.f32s("state_painter", &[_]f32{ m.env.painter.t, m.env.painter.last_value, m.env.painter.start });
            }
            {
                // examples/modules.zig:250-289
                var nb: [64]u8 = undefined;
                const name = try std.fmt.bufPrint(&nb, "hsq_{d}_{d}", .{ ki, fi });
                var out: [2 * F]f32 = undefined;
                var t0: [F]f32 = undefined;
                var t1: [F]f32 = undefined;
                fill(&out, 2700 + ki * 4 + fi, -1.0, 1.0);
                var o = try Out.open(dir, name);
                defer o.close();
                try o.f32s("out0", &out);
                try o.f32s("params", &[_]f32{ SR, freq, 0.0 });
                var m = ex.HardSquareInstrument.init();
                for (script) |st| {
                    const base: usize = if (st.s >= F) F else 0;
                    m.paint(zang.Span.init(st.s - base, st.e - base), .{out[base .. base + F]}, .{ &t0, &t1 }, st.nic, .{ .sample_rate = SR, .freq = freq, .note_on = st.on });
                }
                try o.f32s("out", &out);
                try o.put("state_cnt", u32, &[_]u32{m.osc.cnt});
            }
        }
    }
}

fn basicsAndMixdown(dir: std.fs.Dir) !void {
    var a: [F]f32 = undefined;
    var b: [F]f32 = undefined;
    var d: [F]f32 = undefined;
    fill(&a, 2300, -2.0, 2.0);
    fill(&b, 2301, -2.0, 2.0);
    fill(&d, 2302, -2.0, 2.0);
    {
        var o = try Out.open(dir, "basics");
        defer o.close();
        try o.f32s("a", &a);
        try o.f32s("b", &b);
        try o.f32s("dest0", &d);
        const span = zang.Span.init(100, 900);
        var x = d;
        zang.multiply(span, &x, &a, &b); // dest += a * b: the product is rounded before the add (no FMA)
        try o.f32s("multiply", &x);
        x = d;
        zang.add(span, &x, &a, &b);
        try o.f32s("add", &x);
        x = d;
        zang.addScalar(span, &x, &a, 0.37);
        try o.f32s("addScalar", &x);
        x = d;
        zang.multiplyScalar(span, &x, &a, 0.37);
        try o.f32s("multiplyScalar", &x);
        x = d;
        zang.multiplyWith(span, &x, &a);
        try o.f32s("multiplyWith", &x);
        x = d;
        zang.multiplyWithScalar(span, &x, 0.37);
        try o.f32s("multiplyWithScalar", &x);
        x = d;
        zang.addInto(span, &x, &a);
        try o.f32s("addInto", &x);
    }
    {
        var o = try Out.open(dir, "mixdown");
        defer o.close();
        var mix: [F]f32 = undefined;
        fill(&mix, 2400, -6.0, 6.0); // beyond +-1 / vol: the clamp (mixdown.zig:37-56)
        mix[3] = std.math.nan(f32);
        mix[4] = std.math.inf(f32);
        mix[5] = -std.math.inf(f32);
        try o.f32s("mix", &mix);
        var s16: [F * 2 * 2]u8 = [1]u8{0} ** (F * 2 * 2);
        zang.mixDown(&s16, &mix, .signed16_lsb, 2, 1, 0.25);
        try o.put("s16_2ch_ch1", u8, &s16);
        var s8: [F]u8 = [1]u8{0} ** F;
        zang.mixDown(&s8, &mix, .signed8, 1, 0, 0.25);
        try o.put("s8_1ch", u8, &s8);
    }
}

// The arithmetic that lives in Zig's std / compiler-rt rather than in the reference (SURVEY.md 8c): pinned directly.
fn mathProbes(dir: std.fs.Dir) !void {
    var o = try Out.open(dir, "math");
    defer o.close();
    var x: [4096]f32 = undefined;
    var y: [4096]f32 = undefined;
    fill(&x, 2500, -40.0, 40.0); // SineOsc.zig:5 argument range in practice: (t + phase) * pi * 2
    try o.f32s("sin_x", &x);
    for (x, 0..) |v, i| y[i] = std.math.sin(v);
    try o.f32s("sin", &y);
    for (x, 0..) |v, i| y[i] = std.math.cos(v * 0.08); // Filter.zig:21: pi * f / sr in [0, pi]
    try o.f32s("cos_of_0p08x", &y);
    for (x, 0..) |v, i| y[i] = std.math.atan(v); // Distortion.zig:45,50
    try o.f32s("atan", &y);
    for (x, 0..) |v, i| y[i] = std.math.pow(f32, 2.0, v * 0.2); // Distortion.zig:41: pow(2, ingain * 8 - 2)
    try o.f32s("pow2_of_0p2x", &y);
    // Noise.zig:51's expression over Xoshiro256 seeds 100..107, 512 draws each (does not touch Noise's seed counter)
    var r: [8 * 512]f32 = undefined;
    var k: u64 = 0;
    while (k < 8) : (k += 1) {
        var prng = std.rand.DefaultPrng.init(100 + k);
        const rnd = prng.random();
        var i: usize = 0;
        while (i < 512) : (i += 1) r[k * 512 + i] = rnd.float(f32) * 2.0 - 1.0;
    }
    try o.f32s("white_seeds100to107_x512", &r);
}

// The three places where a restatement of Zig's std could differ silently (VERDICT r3 item 8), pinned by name:
//   * @sin / @cos at and around multiples of pi/2 (SineOsc.zig:5): the reduction's quadrant decisions and its smallest results;
//   * std.math.pow(f32, 2.0, y) at NON-integer y over Distortion.zig:41's range (y = ingain * 8 - 2) and a hair off the integers;
//   * Random.float(f32)'s rare paths (Noise.zig:51): draws with 32..40 leading zeros (the general exponent form) and with 41 or
//     more (a second draw), reached through crafted Xoshiro256 states {0, a, b, 1 << k}: next() = rotl(s0 + s3, 23) + s0.
fn mathProbes2(dir: std.fs.Dir) !void {
    var o = try Out.open(dir, "math2");
    defer o.close();
    var x: [4096]f32 = undefined;
    var y: [4096]f32 = undefined;
    // 455 multiples of pi/2, each at -4 .. +4 ulps (4095 values; the last slot is 0)
    var i: usize = 0;
    var k: u32 = 1;
    while (k <= 455) : (k += 1) {
        const base: f32 = @floatCast(@as(f64, @floatFromInt(k)) * 1.5707963267948966);
        const bits: u32 = @bitCast(base);
        var d: i32 = -4;
        while (d <= 4) : (d += 1) {
            x[i] = @bitCast(bits +% @as(u32, @bitCast(d)));
            i += 1;
        }
    }
    x[4095] = 0.0;
    try o.f32s("sin_pio2_x", &x);
    for (x, 0..) |v, j| y[j] = std.math.sin(v);
    try o.f32s("sin_pio2", &y);
    for (x, 0..) |v, j| y[j] = std.math.cos(v);
    try o.f32s("cos_pio2", &y);
    // exponents of pow(2, .): Distortion's range, and the integers -2 .. 6 at +- 2^-20 .. 2^-17 in the first 72 slots
    fill(&x, 2600, -2.5, 6.5);
    var n: usize = 0;
    var e: i32 = -2;
    while (e <= 6) : (e += 1) {
        var q: u5 = 0;
        while (q < 4) : (q += 1) {
            const off: f32 = 1.0 / @as(f32, @floatFromInt(@as(u32, 1) << (17 + q)));
            x[n] = @as(f32, @floatFromInt(e)) + off;
            x[n + 1] = @as(f32, @floatFromInt(e)) - off;
            n += 2;
        }
    }
    try o.f32s("pow2_y", &x);
    for (x, 0..) |v, j| y[j] = std.math.pow(f32, 2.0, v);
    try o.f32s("pow2", &y);
    // Random.float's rare paths: 10 values of s3 x 4 (a, b) pairs, 4 draws each
    var states: [40 * 4]u64 = undefined;
    var white: [40 * 4]f32 = undefined;
    var st: u64 = 2700;
    var c: usize = 0;
    while (c < 40) : (c += 1) {
        const kk: u6 = @intCast(c % 10);
        const s3: u64 = if (kk < 9) (@as(u64, 1) << kk) else (@as(u64, 1) << 41);
        const a = splitmix(&st) | 1;
        const b = splitmix(&st) | 1;
        states[c * 4 + 0] = 0;
        states[c * 4 + 1] = a;
        states[c * 4 + 2] = b;
        states[c * 4 + 3] = s3;
        var prng = std.rand.DefaultPrng.init(0);
        prng.s = .{ 0, a, b, s3 };
        const rnd = prng.random();
        var d: usize = 0;
        while (d < 4) : (d += 1) white[c * 4 + d] = rnd.float(f32) * 2.0 - 1.0;
    }
    try o.put("rare_states", u64, &states);
    try o.f32s("rare_white", &white);
}

pub fn main() !void {
    var gpa = std.heap.GeneralPurposeAllocator(.{}){};
    defer _ = gpa.deinit();
    const args = try std.process.argsAlloc(gpa.allocator());
    defer std.process.argsFree(gpa.allocator(), args);
    const path = if (args.len > 1) args[1] else "tests/golden/zig";
    try std.fs.cwd().makePath(path);
    var dir = try std.fs.cwd().openDir(path, .{});
    defer dir.close();
    try noise(dir); // FIRST: Noise seeds come from a process-wide counter
    try sineosc(dir);
    try oscillators(dir);
    try envelope(dir);
    try gate(dir);
    try filter(dir);
    try decimator(dir);
    try distortion(dir);
    try sampler(dir);
    try instruments(dir);
    try basicsAndMixdown(dir);
    try mathProbes(dir);
    try mathProbes2(dir);
    std.debug.print("wrote vector files to {s}\n", .{path});
}
