// zscript_front.hip -- zangscript tokenizer, parser and codegen (host C++; see zscript.hpp).
// Restates src/zangscript/tokenize.zig, parse.zig, codegen.zig: same grammar, scoping, type rules, error
// messages, and the same instruction list / temp allocation, so that both backends print what the
// reference's `generateZig` would.
#include "zscript.hpp"

#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace zs {

// ------------------------------------------------------------------ errors (fail.zig:47-111)
void fail(const Source &src, const SourceRange &sr, const std::string &message) {
    const std::string &c = src.contents;
    size_t start = sr.loc0.index;
    while (start > 0 && c[start - 1] != '\n') start--;
    size_t end = sr.loc0.index;
    while (end < c.size() && c[end] != '\n' && c[end] != '\r') end++;
    std::string out = src.filename + ":" + std::to_string(sr.loc0.line + 1) + ":" + std::to_string(sr.loc0.index - start + 1) + ": " + message;
    if (sr.loc0.index != sr.loc1.index) {
        const size_t stop = end < sr.loc1.index ? end : sr.loc1.index;
        out += "\n\n" + c.substr(start, end - start) + "\n" + std::string(sr.loc0.index - start, ' ') + std::string(stop - sr.loc0.index, '^');
    }
    throw ScriptError{out};
}

// ------------------------------------------------------------------ builtins (builtins.zig:153-185)
static const BuiltinEnum kPaintCurve{"PaintCurve", "zang.PaintCurve", {{"instantaneous", false}, {"linear", true}, {"squared", true}, {"cubed", true}}};
static const BuiltinEnum kInterpolation{"InterpolationFunction", "mod.Curve.InterpolationFunction", {{"linear", false}, {"smoothstep", false}}};
static const BuiltinEnum kDistortionType{"DistortionType", "mod.Distortion.Type", {{"overdrive", false}, {"clip", false}}};
static const BuiltinEnum kFilterType{"FilterType", "mod.Filter.Type", {{"bypass", false}, {"low_pass", false}, {"band_pass", false},
                                                                         {"high_pass", false}, {"notch", false}, {"all_pass", false}}};
static const BuiltinEnum kNoiseColor{"NoiseColor", "mod.Noise.Color", {{"white", false}, {"pink", false}}};

static ParamType pt(PK k) { return ParamType{k, nullptr}; }
static ParamType one_of(const BuiltinEnum &e) { return ParamType{PK::one_of, &e}; }

const Package &zang_builtin_package() {
    static const Package p{"zang", "zang", {}, {&kPaintCurve}};
    return p;
}
const Package &modules_builtin_package() {
    // Params in the modules' declaration order (src/modules/*.zig `pub const Params`); the list and its order
    // are builtins.zig:161-176 (Sampler is commented out there too)
    static const Package p{"mod", "modules", {
        {"Curve", {{"sample_rate", pt(PK::constant)}, {"function", one_of(kInterpolation)}, {"curve", pt(PK::curve)}}},
        {"Cycle", {{"sample_rate", pt(PK::constant)}, {"speed", pt(PK::constant_or_buffer)}}},
        {"Decimator", {{"sample_rate", pt(PK::constant)}, {"input", pt(PK::buffer)}, {"fake_sample_rate", pt(PK::constant)}}},
        {"Distortion", {{"input", pt(PK::buffer)}, {"type", one_of(kDistortionType)}, {"ingain", pt(PK::constant)}, {"outgain", pt(PK::constant)}, {"offset", pt(PK::constant)}}},
        {"Envelope", {{"sample_rate", pt(PK::constant)}, {"attack", one_of(kPaintCurve)}, {"decay", one_of(kPaintCurve)}, {"release", one_of(kPaintCurve)},
                      {"sustain_volume", pt(PK::constant)}, {"note_on", pt(PK::boolean)}}},
        {"Filter", {{"input", pt(PK::buffer)}, {"type", one_of(kFilterType)}, {"cutoff", pt(PK::constant_or_buffer)}, {"res", pt(PK::constant_or_buffer)}}},
        {"Gate", {{"note_on", pt(PK::boolean)}}},
        {"Noise", {{"color", one_of(kNoiseColor)}}},
        {"Portamento", {{"sample_rate", pt(PK::constant)}, {"curve", one_of(kPaintCurve)}, {"goal", pt(PK::constant)}, {"note_on", pt(PK::boolean)}, {"prev_note_on", pt(PK::boolean)}}},
        {"PulseOsc", {{"sample_rate", pt(PK::constant)}, {"freq", pt(PK::constant_or_buffer)}, {"color", pt(PK::constant)}}},
        {"SineOsc", {{"sample_rate", pt(PK::constant)}, {"freq", pt(PK::constant_or_buffer)}, {"phase", pt(PK::constant_or_buffer)}}},
        {"TriSawOsc", {{"sample_rate", pt(PK::constant)}, {"freq", pt(PK::constant_or_buffer)}, {"color", pt(PK::constant)}}},
    }, {&kInterpolation, &kDistortionType, &kFilterType, &kNoiseColor}};
    return p;
}

// ------------------------------------------------------------------ tokenizer (tokenize.zig:40-223)
namespace {

struct SymbolText { TT tt; const char *text; };
const SymbolText kSymbols[] = {{TT::sym_asterisk, "*"}, {TT::sym_colon, ":"}, {TT::sym_comma, ","}, {TT::sym_equals, "="},
                               {TT::sym_left_paren, "("}, {TT::sym_minus, "-"}, {TT::sym_plus, "+"}, {TT::sym_right_paren, ")"}, {TT::sym_slash, "/"}};
const SymbolText kKeywords[] = {{TT::kw_begin, "begin"}, {TT::kw_defcurve, "defcurve"}, {TT::kw_defmodule, "defmodule"}, {TT::kw_deftrack, "deftrack"},
                                {TT::kw_delay, "delay"}, {TT::kw_end, "end"}, {TT::kw_false, "false"}, {TT::kw_feedback, "feedback"},
                                {TT::kw_from, "from"}, {TT::kw_out, "out"}, {TT::kw_true, "true"}};

bool is_head(char ch) { return (ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z'); }       // leading underscore is not allowed
bool is_tail(char ch) { return is_head(ch) || (ch >= '0' && ch <= '9') || ch == '_'; }

class Tokenizer {
public:
    const Source &src;
    Loc loc;
    explicit Tokenizer(const Source &s) : src(s) {}

    Token next() {
        const std::string &c = src.contents;
        Loc l = loc;
        struct Commit { Tokenizer *t; Loc *l; ~Commit() { t->loc = *l; } } commit{this, &l};
        for (;;) {
            while (l.index < c.size() && (c[l.index] == ' ' || c[l.index] == '\t' || c[l.index] == '\r' || c[l.index] == '\n')) {
                if (c[l.index] == '\r') {
                    l.index++;
                    if (l.index == c.size() || c[l.index] != '\n') { l.line++; continue; }     // a lone CR ends a line
                }
                if (c[l.index] == '\n') l.line++;
                l.index++;
            }
            if (l.index + 2 < c.size() && c[l.index] == '/' && c[l.index + 1] == '/') {
                while (l.index < c.size() && c[l.index] != '\r' && c[l.index] != '\n') l.index++;
                continue;
            }
            if (l.index == c.size()) return Token{TT::end_of_file, SourceRange{l, l}, 0.0f};
            const Loc start = l;
            for (const SymbolText &s : kSymbols) {
                const size_t n = strlen(s.text);
                if (c.compare(l.index, n, s.text) == 0) { l.index += (uint32_t)n; return Token{s.tt, SourceRange{start, l}, 0.0f}; }
            }
            if (c[l.index] == '.') {
                l.index++;
                const Loc start2 = l;
                if (l.index == c.size() || !is_head(c[l.index])) fail(src, SourceRange{start, start2}, "dot must be followed by an identifier");
                l.index++;
                while (l.index < c.size() && is_tail(c[l.index])) l.index++;
                return Token{TT::enum_value, SourceRange{start2, l}, 0.0f};
            }
            if (c[l.index] >= '0' && c[l.index] <= '9') {
                uint32_t j = l.index + 1;
                while (j < c.size() && ((c[j] >= '0' && c[j] <= '9') || c[j] == '.')) j++;
                const std::string text = c.substr(l.index, j - l.index);
                l.index = j;
                size_t dots = 0;
                for (char ch : text) dots += ch == '.';
                if (dots > 1) fail(src, SourceRange{start, l}, "malformatted number");
                return Token{TT::number, SourceRange{start, l}, strtof(text.c_str(), nullptr)};
            }
            if (is_head(c[l.index])) {
                l.index++;
                while (l.index < c.size() && is_tail(c[l.index])) l.index++;
                const std::string text = c.substr(start.index, l.index - start.index);
                TT tt = TT::name;
                for (const SymbolText &k : kKeywords) if (text == k.text) tt = k.tt;
                return Token{tt, SourceRange{start, l}, 0.0f};
            }
            l.index++;
            return Token{TT::illegal, SourceRange{start, l}, 0.0f};
        }
    }
    Token peek() {
        const Loc saved = loc;
        Token t = next();
        loc = saved;
        return t;
    }
    [[noreturn]] void fail_expected(const std::string &desc, const Token &found) {            // tokenize.zig:140-146
        if (found.tt == TT::end_of_file) fail(src, found.sr, "expected " + desc + ", found end of file");
        fail(src, found.sr, "expected " + desc + ", found `" + src.text(found.sr) + "`");
    }
    Token expect_next(TT tt) {                                                                  // :149-158
        Token t = next();
        if (t.tt == tt) return t;
        const char *text = "?";
        for (const SymbolText &s : kSymbols) if (s.tt == tt) text = s.text;
        for (const SymbolText &k : kKeywords) if (k.tt == tt) text = k.text;
        fail_expected(std::string("`") + text + "`", t);
    }
};

// ------------------------------------------------------------------ parser (parse.zig)
const char *kReserved[] = {"abs", "cos", "max", "min", "pi", "pow", "sample_rate", "sin", "sqrt"};     // :185-195
bool is_reserved(const std::string &n) { for (const char *r : kReserved) if (n == r) return true; return false; }

struct ModuleState { std::vector<ModuleParam> params; std::vector<std::string> locals; };
struct PC { ModuleState *ms = nullptr; Scope *scope = nullptr; bool module() const { return ms != nullptr; } };     // ParseContext

class Parser {
public:
    const Source &src;
    Tokenizer tok;
    ParseResult &out;
    std::vector<const BuiltinEnum *> enums;

    Parser(const Source &s, const std::vector<const Package *> &packages, ParseResult &o) : src(s), tok(s), out(o) {
        for (const Package *pkg : packages) {                                                    // parse.zig:766-787
            for (const BuiltinEnum *e : pkg->enums) enums.push_back(e);
            for (const BuiltinModule &b : pkg->builtins) {
                Module m;
                m.params = b.params; m.builtin = true; m.builtin_name = b.name; m.zig_package_name = pkg->zig_package_name;
                const size_t idx = out.modules.size();
                out.modules.push_back(m);
                ExprP e = std::make_shared<Expr>();
                e->kind = EK::literal_module; e->index = idx;
                out.globals.push_back(Global{b.name, e});
            }
        }
    }
    std::string text(const SourceRange &sr) const { return src.text(sr); }
    ExprP mk(EK kind, Loc loc0) {                     // createExpr: ends at the tokenizer's position (:452-456)
        ExprP e = std::make_shared<Expr>();
        e->kind = kind; e->sr = SourceRange{loc0, tok.loc};
        return e;
    }
    Scope *new_scope(Scope *parent) {
        out.scopes.emplace_back(new Scope());
        out.scopes.back()->parent = parent;
        return out.scopes.back().get();
    }

    size_t define_curve() {                                                                      // :197-231
        Curve curve;
        bool have = false;
        float last_t = 0;
        for (;;) {
            Token t = tok.next();
            if (t.tt == TT::kw_end) break;
            if (t.tt != TT::number) tok.fail_expected("number or `end`", t);
            if (have && t.number <= last_t) fail(src, t.sr, "time value must be greater than the previous time value");
            have = true; last_t = t.number;
            Token v = tok.next();
            if (v.tt != TT::number) tok.fail_expected("number", v);
            curve.points.push_back({NumberLiteral{t.number, text(t.sr)}, NumberLiteral{v.number, text(v.sr)}});
        }
        out.curves.push_back(curve);
        return out.curves.size() - 1;
    }
    ParamType expect_param_type(bool for_track) {                                                // :233-256
        Token t = tok.next();
        if (t.tt != TT::name) tok.fail_expected("param type", t);
        const std::string name = text(t.sr);
        ParamType p;
        if (name == "boolean") p = pt(PK::boolean);
        else if (name == "constant") p = pt(PK::constant);
        else if (name == "waveform") p = pt(PK::buffer);
        else if (name == "cob") p = pt(PK::constant_or_buffer);
        else if (name == "curve") p = pt(PK::curve);
        else {
            const BuiltinEnum *found = nullptr;
            for (const BuiltinEnum *e : enums) if (e->name == name) { found = e; break; }
            if (!found) tok.fail_expected("param type", t);
            p = one_of(*found);
        }
        if (for_track && (p.kind == PK::buffer || p.kind == PK::constant_or_buffer)) fail(src, t.sr, "track param cannot be cob or waveform");
        return p;
    }
    void parse_param_declarations(std::vector<ModuleParam> &params, bool for_track) {            // :258-288
        for (;;) {
            Token t = tok.next();
            if (t.tt == TT::kw_begin) return;
            if (t.tt != TT::name) tok.fail_expected("param declaration or `begin`", t);
            const std::string name = text(t.sr);
            if (is_reserved(name)) fail(src, t.sr, "`" + name + "` is a reserved name");
            for (const ModuleParam &p : params) if (p.name == name) fail(src, t.sr, "redeclaration of param `" + name + "`");
            tok.expect_next(TT::sym_colon);
            const ParamType p = expect_param_type(for_track);
            tok.expect_next(TT::sym_comma);
            params.push_back(ModuleParam{name, p});
        }
    }
    size_t define_track() {                                                                      // :290-326
        Track track;
        parse_param_declarations(track.params, true);
        bool have = false;
        float last_t = 0;
        for (;;) {
            Token t = tok.next();
            if (t.tt == TT::kw_end) break;
            if (t.tt != TT::number) tok.fail_expected("number or `end`", t);
            if (have && t.number <= last_t) fail(src, t.sr, "time value must be greater than the previous time value");
            have = true; last_t = t.number;
            const Loc loc0 = tok.loc;
            TrackNote note;
            note.args = parse_call_args(PC{});
            note.t = NumberLiteral{t.number, text(t.sr)};
            note.args_sr = SourceRange{loc0, tok.loc};
            track.notes.push_back(note);
        }
        out.tracks.push_back(track);
        return out.tracks.size() - 1;
    }
    size_t define_module() {                                                                     // :328-359
        ModuleState ms;
        ms.params.push_back(ModuleParam{"sample_rate", pt(PK::constant)});                        // implicitly declared
        parse_param_declarations(ms.params, false);
        Scope *scope = parse_statements(ms, nullptr);
        Module m;
        m.params = ms.params; m.scope = scope; m.locals = ms.locals;
        out.modules.push_back(m);
        return out.modules.size() - 1;
    }

    std::vector<CallArg> parse_call_args(PC pc) {                                                // :366-410
        tok.expect_next(TT::sym_left_paren);
        std::vector<CallArg> args;
        Token t = tok.next();
        while (t.tt != TT::sym_right_paren) {
            if (!args.empty()) {
                if (t.tt != TT::sym_comma) tok.fail_expected("`,` or `)`", t);
                t = tok.next();
            }
            if (t.tt != TT::name) tok.fail_expected("callee param name", t);
            const std::string name = text(t.sr);
            Token eq = tok.next();
            if (eq.tt == TT::sym_equals) {
                ExprP v = expect_expression(pc, 0);
                args.push_back(CallArg{name, t, v});
                t = tok.next();
            } else if (pc.module()) {                           // shorthand: `val` expands to `val=val`
                args.push_back(CallArg{name, t, resolve_name(pc, t)});
                t = eq;
            } else {
                tok.fail_expected("`=`", eq);                   // (the reference loops here; reported instead)
            }
        }
        return args;
    }
    ExprP resolve_name(PC pc, const Token &t) {                                                  // :458-493
        if (pc.module()) {
            const std::string name = text(t.sr);
            for (Scope *sc = pc.scope; sc; sc = sc->parent)
                for (size_t i = sc->statements.size(); i > 0; i--) {                             // later declarations shadow earlier ones
                    const Statement &st = sc->statements[i - 1];
                    if (st.kind == SK::let_assignment && pc.ms->locals[st.local_index] == name) {
                        ExprP e = std::make_shared<Expr>();
                        e->kind = EK::local; e->sr = t.sr; e->index = st.local_index;
                        return e;
                    }
                }
        }
        ExprP e = std::make_shared<Expr>();                     // a param or a global: resolved in codegen
        e->kind = EK::name; e->sr = t.sr; e->token = t;
        return e;
    }
    ExprP expect_expression(PC pc, size_t priority) {                                            // :520-565
        bool negate = false;
        if (tok.peek().tt == TT::sym_minus) { tok.next(); negate = true; }
        ExprP a = expect_term(pc);
        const Loc loc0 = a->sr.loc0;
        if (tok.peek().tt == TT::sym_left_paren) {
            if (!pc.module()) fail(src, a->sr, "not a function");
            std::vector<CallArg> args = parse_call_args(pc);
            ExprP c = mk(EK::call, loc0);
            c->a = a; c->args = args;
            a = c;
        }
        if (negate) { ExprP n = mk(EK::un_arith, loc0); n->op = "neg"; n->a = a; a = n; }
        static const struct { TT sym; size_t prio; const char *op; } ops[] = {{TT::sym_plus, 1, "add"}, {TT::sym_minus, 1, "sub"},
                                                                            {TT::sym_asterisk, 2, "mul"}, {TT::sym_slash, 2, "div"}};
        for (;;) {
            const Token t = tok.peek();
            bool matched = false;
            for (const auto &bo : ops)
                if (t.tt == bo.sym && priority < bo.prio) {
                    tok.next();
                    ExprP b = expect_expression(pc, bo.prio);
                    ExprP e = mk(EK::bin_arith, loc0);
                    e->op = bo.op; e->a = a; e->b = b;
                    a = e;
                    matched = true;
                    break;
                }
            if (!matched) return a;
        }
    }
    ExprP unary(PC pc, Loc loc0, const char *op) {
        tok.expect_next(TT::sym_left_paren);
        ExprP a = expect_expression(pc, 0);
        tok.expect_next(TT::sym_right_paren);
        ExprP e = mk(EK::un_arith, loc0);
        e->op = op; e->a = a;
        return e;
    }
    ExprP binary(PC pc, Loc loc0, const char *op) {
        tok.expect_next(TT::sym_left_paren);
        ExprP a = expect_expression(pc, 0);
        tok.expect_next(TT::sym_comma);
        ExprP b = expect_expression(pc, 0);
        tok.expect_next(TT::sym_right_paren);
        ExprP e = mk(EK::bin_arith, loc0);
        e->op = op; e->a = a; e->b = b;
        return e;
    }
    ExprP expect_term(PC pc) {                                                                   // :584-690
        const Token t = tok.next();
        const Loc loc0 = t.sr.loc0;
        switch (t.tt) {
        case TT::sym_left_paren: { ExprP a = expect_expression(pc, 0); tok.expect_next(TT::sym_right_paren); return a; }
        case TT::kw_defmodule: { const size_t i = define_module(); ExprP e = mk(EK::literal_module, loc0); e->index = i; return e; }
        case TT::kw_defcurve: { const size_t i = define_curve(); ExprP e = mk(EK::literal_curve, loc0); e->index = i; return e; }
        case TT::kw_deftrack: { const size_t i = define_track(); ExprP e = mk(EK::literal_track, loc0); e->index = i; return e; }
        case TT::kw_from: {
            if (!pc.module()) fail(src, t.sr, "cannot call track outside of module context");
            ExprP track = expect_expression(pc, 0);                                              // parseTrackCall :412-424
            tok.expect_next(TT::sym_comma);
            ExprP speed = expect_expression(pc, 0);
            tok.expect_next(TT::kw_begin);
            Scope *inner = parse_statements(*pc.ms, pc.scope);
            ExprP e = mk(EK::track_call, loc0);
            e->a = track; e->b = speed; e->scope = inner;
            return e;
        }
        case TT::name: {
            const std::string s = text(t.sr);
            if (s == "abs" || s == "cos" || s == "sin" || s == "sqrt") return unary(pc, loc0, s.c_str());
            if (s == "max" || s == "min" || s == "pow") return binary(pc, loc0, s.c_str());
            if (s == "pi") { ExprP e = mk(EK::literal_number, loc0); e->num = NumberLiteral{(float)M_PI, "std.math.pi"}; return e; }
            ExprP r = resolve_name(pc, t);
            r->sr = SourceRange{loc0, tok.loc};
            return r;
        }
        case TT::kw_false: { ExprP e = mk(EK::literal_boolean, loc0); e->bval = false; return e; }
        case TT::kw_true: { ExprP e = mk(EK::literal_boolean, loc0); e->bval = true; return e; }
        case TT::number: { ExprP e = mk(EK::literal_number, loc0); e->num = NumberLiteral{t.number, text(t.sr)}; return e; }
        case TT::enum_value: {
            const std::string label = text(t.sr);
            if (tok.peek().tt == TT::sym_left_paren) {
                tok.next();
                ExprP payload = expect_expression(pc, 0);
                tok.expect_next(TT::sym_right_paren);
                ExprP e = mk(EK::literal_enum_value, loc0);
                e->label = label; e->a = payload;
                return e;
            }
            ExprP e = std::make_shared<Expr>();
            e->kind = EK::literal_enum_value; e->sr = t.sr; e->label = label;
            return e;
        }
        case TT::kw_delay: {
            if (!pc.module()) fail(src, t.sr, "cannot use delay outside of module context");
            const Token n = tok.next();                                                          // parseDelay :426-446
            if (n.tt != TT::number) tok.fail_expected("number", n);
            const std::string s = text(n.sr);
            for (char ch : s) if (ch < '0' || ch > '9') fail(src, n.sr, "malformatted integer");
            tok.expect_next(TT::kw_begin);
            Scope *inner = parse_statements(*pc.ms, pc.scope);
            ExprP e = mk(EK::delay, loc0);
            e->index = (size_t)strtoull(s.c_str(), nullptr, 10); e->scope = inner;
            return e;
        }
        case TT::kw_feedback:
            if (!pc.module()) fail(src, t.sr, "cannot use feedback outside of module context");
            return mk(EK::feedback, loc0);
        default: tok.fail_expected("expression", t);
        }
    }
    Scope *parse_statements(ModuleState &ms, Scope *parent) {                                    // :734-764
        Scope *scope = new_scope(parent);
        const PC pc{&ms, scope};
        for (;;) {
            const Token t = tok.next();
            if (t.tt == TT::kw_end) return scope;
            if (t.tt == TT::name) {                                                              // parseLocalDecl :692-713
                const std::string name = text(t.sr);
                tok.expect_next(TT::sym_equals);
                if (is_reserved(name)) fail(src, t.sr, "`" + name + "` is a reserved name");
                ExprP e = expect_expression(pc, 0);              // the new local is not yet visible to its own initialiser
                ms.locals.push_back(name);
                scope->statements.push_back(Statement{SK::let_assignment, e, ms.locals.size() - 1});
            } else if (t.tt == TT::kw_out) {
                scope->statements.push_back(Statement{SK::output, expect_expression(pc, 0), 0});
            } else if (t.tt == TT::kw_feedback) {
                scope->statements.push_back(Statement{SK::feedback, expect_expression(pc, 0), 0});
            } else {
                tok.fail_expected("local declaration, `out`, `feedback` or `end`", t);
            }
        }
    }
    void run() {                                                                                 // :789-796
        for (;;) {
            const Token t = tok.next();
            if (t.tt == TT::end_of_file) return;
            if (t.tt != TT::name) tok.fail_expected("declaration or end of file", t);
            const std::string name = text(t.sr);                                                 // parseGlobalDecl :715-732
            tok.expect_next(TT::sym_equals);
            if (is_reserved(name)) fail(src, t.sr, "`" + name + "` is a reserved name");
            for (const Global &g : out.globals) if (g.name == name) fail(src, t.sr, "redeclaration of global `" + name + "`");
            ExprP v = expect_expression(PC{}, 0);
            out.globals.push_back(Global{name, v});
        }
    }
};

}  // namespace

void parse(const Source &src, const std::vector<const Package *> &packages, ParseResult &out) {
    Parser p(src, packages, out);
    p.run();
}

// ------------------------------------------------------------------ codegen (codegen.zig)
namespace {

struct Temps {                                        // TempManager (:175-223)
    bool reuse;
    std::vector<bool> claimed;
    size_t claim() {
        if (reuse)
            for (size_t i = 0; i < claimed.size(); i++)
                if (!claimed[i]) { claimed[i] = true; return i; }
        claimed.push_back(true);
        return claimed.size() - 1;
    }
    void release(size_t i) { claimed[i] = false; }
};

struct CMS {                                          // CodegenModuleState
    size_t module_index;
    std::vector<Instr> instructions;
    Temps temp_buffers{true, {}}, temp_floats{false, {}};            // floats become `const` in Zig: never reused
    std::vector<std::pair<bool, Res>> local_results;
    std::vector<size_t> fields, delays, triggers, note_trackers;
    bool in_delay = false, in_track = false;
    size_t delay_feedback_temp = 0, track_index = 0;
    std::vector<Instr> *nested = nullptr;             // instruction list of the running delay / track call
};

class CodeGen {
public:
    CompiledScript &cs;
    const Source &src;
    std::vector<Global> &globals;
    std::vector<Module> &modules;
    std::vector<Track> &tracks;
    std::vector<std::pair<bool, Res>> global_results;
    std::vector<bool> global_visited, track_done, module_done, module_visiting;

    explicit CodeGen(CompiledScript &c) : cs(c), src(c.source), globals(c.pr.globals), modules(c.pr.modules), tracks(c.pr.tracks) {
        global_results.resize(globals.size());
        global_visited.assign(globals.size(), false);
        track_done.assign(tracks.size(), false);
        module_done.assign(modules.size(), false);
        module_visiting.assign(modules.size(), false);
        cs.track_results.resize(tracks.size());
        cs.module_results.resize(modules.size());
    }

    // ---- result classification (:243-377)
    const ParamType *param_type(CMS *cms, const Res &r) const {
        if (r.kind == RK::self_param) return &modules[cms->module_index].params[r.index].type;
        if (r.kind == RK::track_param) return &tracks[r.track_index].params[r.index].type;
        return nullptr;
    }
    bool is_boolean(CMS *cms, const Res &r) const { const ParamType *p = param_type(cms, r); return r.kind == RK::literal_boolean || (p && p->kind == PK::boolean); }
    bool is_float(CMS *cms, const Res &r) const { const ParamType *p = param_type(cms, r); return r.kind == RK::temp_float || r.kind == RK::literal_number || (p && p->kind == PK::constant); }
    bool is_buffer(CMS *cms, const Res &r) const { const ParamType *p = param_type(cms, r); return r.kind == RK::temp_buffer || (p && p->kind == PK::buffer); }
    bool is_curve(CMS *cms, const Res &r) const { const ParamType *p = param_type(cms, r); return r.kind == RK::literal_curve || (p && p->kind == PK::curve); }
    static bool enum_allows(const std::vector<EnumValue> &allowed, const std::string &label, bool has_float) {
        for (const EnumValue &v : allowed) if (v.label == label) return v.f32_payload == has_float;
        return false;
    }
    bool is_enum_value(CMS *cms, const Res &r, const std::vector<EnumValue> &allowed) const {
        if (r.kind == RK::literal_enum_value) return enum_allows(allowed, r.label, r.payload && is_float(cms, *r.payload));
        const ParamType *p = param_type(cms, r);
        if (p && p->kind == PK::one_of) {                                 // every possible value must be allowed
            for (const EnumValue &v : p->en->values) if (!enum_allows(allowed, v.label, v.f32_payload)) return false;
            return true;
        }
        return false;
    }

    // ---- temps / destinations (:379-424)
    void release(CMS *cms, const Res &r) {
        if (r.kind == RK::temp_buffer && !r.weak) cms->temp_buffers.release(r.index);
        else if (r.kind == RK::temp_float && !r.weak) cms->temp_floats.release(r.index);
        else if (r.kind == RK::literal_enum_value && r.payload) release(cms, *r.payload);
    }
    static Dest request_dest(CMS *cms, const Dest *loc) { return loc ? *loc : Dest{false, cms->temp_buffers.claim()}; }
    static Res commit_dest(const Dest *loc, const Dest &d) {
        Res r;
        if (loc) return r;                                                // nothing
        r.kind = RK::temp_buffer; r.index = d.index;
        return r;
    }
    static void add(CMS *cms, const Instr &i) { (cms->nested ? *cms->nested : cms->instructions).push_back(i); }   // :415-423
    static Res mk(RK k, size_t index = 0) { Res r; r.kind = k; r.index = index; return r; }

    // ---- arithmetic (:438-500)
    Res gen_un_arith(CMS *cms, const SourceRange &sr, const Dest *loc, const std::string &op, const ExprP &ea) {
        const Res ra = gen_expression(cms, ea, nullptr);
        struct Rel { CodeGen *g; CMS *c; const Res *r; ~Rel() { g->release(c, *r); } } rel{this, cms, &ra};
        if (is_float(cms, ra)) {
            const size_t idx = cms->temp_floats.claim();
            Instr i; i.kind = IK::arith_float; i.out_float = idx; i.op = op; i.a = ra;
            add(cms, i);
            return mk(RK::temp_float, idx);
        }
        if (is_buffer(cms, ra)) {
            const Dest d = request_dest(cms, loc);
            Instr i; i.kind = IK::arith_buffer; i.out = d; i.op = op; i.a = ra;
            add(cms, i);
            return commit_dest(loc, d);
        }
        fail(src, sr, "arithmetic can only be performed on numeric types");
    }
    Res gen_bin_arith(CMS *cms, const SourceRange &sr, const Dest *loc, const std::string &op, const ExprP &ea, const ExprP &eb) {
        const Res ra = gen_expression(cms, ea, nullptr);
        struct Rel { CodeGen *g; CMS *c; const Res *r; ~Rel() { g->release(c, *r); } } rel_a{this, cms, &ra};
        const Res rb = gen_expression(cms, eb, nullptr);
        Rel rel_b{this, cms, &rb};                                        // destroyed first: rb released, then ra (Zig defer order)
        const bool fa = is_float(cms, ra), fb = is_float(cms, rb), ba = is_buffer(cms, ra), bb = is_buffer(cms, rb);
        if (fa && fb) {
            const size_t idx = cms->temp_floats.claim();
            Instr i; i.kind = IK::arith_float_float; i.out_float = idx; i.op = op; i.a = ra; i.b = rb;
            add(cms, i);
            return mk(RK::temp_float, idx);
        }
        IK kind;
        if (fa && bb) kind = IK::arith_float_buffer;
        else if (ba && fb) kind = IK::arith_buffer_float;
        else if (ba && bb) kind = IK::arith_buffer_buffer;
        else fail(src, sr, "arithmetic can only be performed on numeric types");
        const Dest d = request_dest(cms, loc);
        Instr i; i.kind = kind; i.out = d; i.op = op; i.a = ra; i.b = rb;
        add(cms, i);
        return commit_dest(loc, d);
    }

    // ---- calls (:502-620)
    Res commit_callee_param(CMS *cms, const SourceRange &sr, const Res &r, const ParamType &p) {
        switch (p.kind) {
        case PK::boolean: if (is_boolean(cms, r)) return r; fail(src, sr, "expected boolean value");
        case PK::buffer:
            if (is_buffer(cms, r)) return r;
            if (is_float(cms, r)) {
                const size_t idx = cms->temp_buffers.claim();
                Instr i; i.kind = IK::float_to_buffer; i.out = Dest{false, idx}; i.src = r;
                add(cms, i);
                return mk(RK::temp_buffer, idx);
            }
            fail(src, sr, "expected buffer value");
        case PK::constant_or_buffer: if (is_buffer(cms, r) || is_float(cms, r)) return r; fail(src, sr, "expected float or buffer value");
        case PK::constant: if (is_float(cms, r)) return r; fail(src, sr, "expected float value");
        case PK::curve: if (is_curve(cms, r)) return r; fail(src, sr, "expected curve value");
        case PK::one_of: {
            if (is_enum_value(cms, r, p.en->values)) return r;
            std::string names;
            for (size_t i = 0; i < p.en->values.size(); i++) {
                if (i) names += ", ";
                names += "'" + p.en->values[i].label + "'" + (p.en->values[i].f32_payload ? "(number)" : "");
            }
            fail(src, sr, "expected one of " + names);
        }
        }
        fail(src, sr, "internal: param type");
    }
    std::vector<Res> gen_args(CMS *cms, const SourceRange &sr, const std::vector<ModuleParam> &params, const std::vector<CallArg> &args) {
        for (const CallArg &a : args) {
            bool found = false;
            for (const ModuleParam &p : params) found |= p.name == a.param_name;
            if (!found) fail(src, a.token.sr, "call target has no param called `" + src.text(a.token.sr) + "`");
        }
        std::vector<Res> results;
        for (const ModuleParam &p : params) {
            const CallArg *arg = nullptr;
            for (const CallArg &a : args) {
                if (a.param_name != p.name) continue;
                if (arg) fail(src, a.token.sr, "param `" + src.text(a.token.sr) + "` provided more than once");
                arg = &a;
            }
            if (cms && !arg && p.name == "sample_rate") {                                        // passed implicitly
                const std::vector<ModuleParam> &self = modules[cms->module_index].params;
                size_t j = 0;
                while (self[j].name != "sample_rate") j++;
                results.push_back(mk(RK::self_param, j));
                continue;
            }
            if (!arg) fail(src, sr, "argument list is missing param `" + p.name + "`");
            const Res r = gen_expression(cms, arg->value, nullptr);
            results.push_back(commit_callee_param(cms, arg->value->sr, r, p.type));
        }
        return results;
    }
    Res gen_call(CMS *cms, const SourceRange &sr, const Dest *loc, const Expr &call) {
        const Res fr = gen_expression(cms, call.a, nullptr);
        if (fr.kind != RK::literal_module) fail(src, call.a->sr, "not a module");
        const size_t callee_index = fr.index, field_index = cms->fields.size();
        cms->fields.push_back(callee_index);
        const std::vector<Res> arg_results = gen_args(cms, sr, modules[callee_index].params, call.args);
        std::vector<size_t> temps;
        for (size_t i = 0; i < cs.module_results[callee_index].num_temps; i++) temps.push_back(cms->temp_buffers.claim());
        const Dest d = request_dest(cms, loc);
        Instr i; i.kind = IK::call; i.out = d; i.field_index = field_index; i.temps = temps; i.args = arg_results;
        add(cms, i);
        const Res result = commit_dest(loc, d);
        for (size_t t : temps) cms->temp_buffers.release(t);
        for (const Res &r : arg_results) release(cms, r);
        return result;
    }
    void gen_inner_statements(CMS *cms, Scope *scope, const Dest &dest, const Dest *feedback_dest) {
        for (const Statement &st : scope->statements) {
            if (st.kind == SK::let_assignment) {
                cms->local_results[st.local_index] = {true, gen_expression(cms, st.expr, nullptr)};
            } else if (st.kind == SK::output) {
                const Res r = gen_expression(cms, st.expr, &dest);
                commit_output(cms, st.expr->sr, r, dest);
                release(cms, r);
            } else {
                if (!feedback_dest) fail(src, st.expr->sr, "`feedback` can only be used within a `delay` operation");
                const Res r = gen_expression(cms, st.expr, feedback_dest);
                commit_output(cms, st.expr->sr, r, *feedback_dest);
                release(cms, r);
            }
        }
    }
    Res gen_track_call(CMS *cms, const SourceRange &sr, const Dest *loc, const Expr &e) {         // :558-626
        if (cms->in_track) fail(src, sr, "you cannot nest track calls");
        if (cms->in_delay) fail(src, sr, "you cannot use a track call inside a delay");
        const Res tr = gen_expression(cms, e.a, nullptr);
        if (tr.kind != RK::literal_track) fail(src, e.a->sr, "not a track");
        const Res speed = gen_expression(cms, e.b, nullptr);
        if (!is_float(cms, speed)) fail(src, e.b->sr, "speed must be a constant value");
        const size_t trigger_index = cms->triggers.size();
        cms->triggers.push_back(tr.index);
        const size_t note_tracker_index = cms->note_trackers.size();
        cms->note_trackers.push_back(tr.index);
        const Dest d = request_dest(cms, loc);
        std::vector<Instr> inner;
        cms->in_track = true; cms->track_index = tr.index; cms->nested = &inner;
        gen_inner_statements(cms, e.scope, d, nullptr);
        cms->in_track = false; cms->nested = nullptr;
        Instr i; i.kind = IK::track_call; i.out = d; i.track_index = tr.index; i.speed = speed; i.trigger_index = trigger_index;
        i.note_tracker_index = note_tracker_index; i.instructions = inner;
        add(cms, i);
        release(cms, speed);
        return commit_dest(loc, d);
    }
    Res gen_delay(CMS *cms, const SourceRange &sr, const Dest *loc, const Expr &e) {              // :628-690
        if (cms->in_delay) fail(src, sr, "you cannot nest delay operations");
        if (cms->in_track) fail(src, sr, "you cannot use a delay inside a track call");
        const size_t delay_index = cms->delays.size();
        cms->delays.push_back(e.index);
        const size_t feedback_temp = cms->temp_buffers.claim();
        const Dest d = request_dest(cms, loc);
        const size_t feedback_out_temp = cms->temp_buffers.claim();
        std::vector<Instr> inner;
        cms->in_delay = true; cms->delay_feedback_temp = feedback_temp; cms->nested = &inner;
        const Dest fb{false, feedback_out_temp};
        gen_inner_statements(cms, e.scope, d, &fb);
        cms->in_delay = false; cms->nested = nullptr;
        Instr i; i.kind = IK::delay; i.out = d; i.delay_index = delay_index; i.feedback_out_temp = feedback_out_temp; i.feedback_temp = feedback_temp;
        i.instructions = inner;
        add(cms, i);
        const Res result = commit_dest(loc, d);
        cms->temp_buffers.release(feedback_out_temp);
        cms->temp_buffers.release(feedback_temp);
        return result;
    }
    void gen_track(size_t ti) {                                                                  // :692-706
        if (track_done[ti]) return;
        track_done[ti] = true;
        for (const TrackNote &n : tracks[ti].notes) cs.track_results[ti].push_back(gen_args(nullptr, n.args_sr, tracks[ti].params, n.args));
    }
    void gen_module(size_t mi, const SourceRange &sr) {                                          // :708-767
        if (module_done[mi]) return;
        // a module that (directly or not) calls itself: the reference recurses until the stack ends
        if (module_visiting[mi]) fail(src, sr, "circular reference in module");
        module_visiting[mi] = true;
        CMS cms;
        cms.module_index = mi;
        cms.local_results.resize(modules[mi].locals.size());
        for (const Statement &st : modules[mi].scope->statements) {
            if (st.kind == SK::let_assignment) {
                cms.local_results[st.local_index] = {true, gen_expression(&cms, st.expr, nullptr)};
            } else if (st.kind == SK::output) {
                const Dest loc{true, 0};
                const Res r = gen_expression(&cms, st.expr, &loc);
                commit_output(&cms, st.expr->sr, r, loc);
                release(&cms, r);
            } else {
                fail(src, st.expr->sr, "`feedback` can only be used within a `delay` operation");
            }
        }
        for (const auto &lr : cms.local_results) if (lr.first) release(&cms, lr.second);
        ModuleResult mr;
        mr.num_temps = cms.temp_buffers.claimed.size(); mr.num_temp_floats = cms.temp_floats.claimed.size();
        mr.fields = cms.fields; mr.delays = cms.delays; mr.note_trackers = cms.note_trackers; mr.triggers = cms.triggers;
        mr.instructions = cms.instructions;
        cs.module_results[mi] = mr;
        module_done[mi] = true;
    }

    // ---- expressions (:775-911); cms == nullptr in the global context
    static Res weaken(Res r) { if (r.kind == RK::temp_buffer || r.kind == RK::temp_float) r.weak = true; return r; }
    Res gen_expression(CMS *cms, const ExprP &ep, const Dest *loc) {
        const Expr &e = *ep;
        switch (e.kind) {
        case EK::literal_boolean: { Res r = mk(RK::literal_boolean); r.bval = e.bval; return r; }
        case EK::literal_number: { Res r = mk(RK::literal_number); r.num = e.num; return r; }
        case EK::literal_enum_value: {
            Res r = mk(RK::literal_enum_value);
            r.label = e.label;
            if (e.a) r.payload = std::make_shared<Res>(gen_expression(cms, e.a, nullptr));
            return r;
        }
        case EK::literal_curve: return mk(RK::literal_curve, e.index);
        case EK::literal_track: gen_track(e.index); return mk(RK::literal_track, e.index);
        case EK::literal_module: if (!modules[e.index].builtin) gen_module(e.index, e.sr); return mk(RK::literal_module, e.index);
        case EK::name: {
            const std::string name = src.text(e.token.sr);
            if (cms) {
                if (cms->in_track)
                    for (size_t pi = 0; pi < tracks[cms->track_index].params.size(); pi++)
                        if (tracks[cms->track_index].params[pi].name == name) { Res r = mk(RK::track_param, pi); r.track_index = cms->track_index; return r; }
                const std::vector<ModuleParam> &params = modules[cms->module_index].params;
                for (size_t pi = 0; pi < params.size(); pi++) {
                    if (params[pi].name != name) continue;
                    if (params[pi].type.kind == PK::constant_or_buffer) {                         // unwrapped into a buffer at once (:843-848)
                        const Dest d = request_dest(cms, loc);
                        Instr i; i.kind = IK::cob_to_buffer; i.out = d; i.in_self_param = pi;
                        add(cms, i);
                        return commit_dest(loc, d);
                    }
                    return mk(RK::self_param, pi);
                }
            }
            size_t gi = 0;
            while (gi < globals.size() && globals[gi].name != name) gi++;
            if (gi == globals.size()) fail(src, e.token.sr, "use of undeclared identifier `" + name + "`");
            if (!global_results[gi].first) {
                if (global_visited[gi]) fail(src, e.token.sr, "circular reference in global");
                global_visited[gi] = true;
                global_results[gi] = {true, gen_expression(nullptr, globals[gi].value, nullptr)};
            }
            return weaken(global_results[gi].second);
        }
        case EK::local: return weaken(cms->local_results[e.index].second);
        case EK::un_arith:
            if (!cms) fail(src, e.sr, "constant arithmetic is not supported");
            return gen_un_arith(cms, e.sr, loc, e.op, e.a);
        case EK::bin_arith:
            if (!cms) fail(src, e.sr, "constant arithmetic is not supported");
            return gen_bin_arith(cms, e.sr, loc, e.op, e.a, e.b);
        case EK::call: return gen_call(cms, e.sr, loc, e);
        case EK::track_call: return gen_track_call(cms, e.sr, loc, e);
        case EK::delay: return gen_delay(cms, e.sr, loc, e);
        case EK::feedback: {
            if (!cms->in_delay) fail(src, e.sr, "`feedback` can only be used within a `delay` operation");
            Res r = mk(RK::temp_buffer, cms->delay_feedback_temp);
            r.weak = true;
            return r;
        }
        }
        fail(src, e.sr, "internal: expression kind");
    }
    void commit_output(CMS *cms, const SourceRange &sr, const Res &r, const Dest &d) {             // :913-960
        Instr i;
        i.out = d; i.src = r;
        switch (r.kind) {
        case RK::nothing: return;
        case RK::temp_buffer: i.kind = IK::copy_buffer; add(cms, i); return;
        case RK::temp_float: case RK::literal_number: i.kind = IK::float_to_buffer; add(cms, i); return;
        case RK::self_param: case RK::track_param: {
            const PK k = param_type(cms, r)->kind;
            if (k == PK::buffer || k == PK::constant_or_buffer) { i.kind = IK::copy_buffer; add(cms, i); return; }
            if (k == PK::constant) { i.kind = IK::float_to_buffer; add(cms, i); return; }
            fail(src, sr, std::string("expected buffer value, found ") + (k == PK::boolean ? "boolean" : k == PK::curve ? "curve" : "enum value"));
        }
        case RK::literal_boolean: fail(src, sr, "expected buffer value, found boolean");
        case RK::literal_enum_value: fail(src, sr, "expected buffer value, found enum value");
        case RK::literal_curve: fail(src, sr, "expected buffer value, found curve");
        case RK::literal_track: fail(src, sr, "expected buffer value, found track");
        case RK::literal_module: fail(src, sr, "expected buffer value, found module");
        }
    }
    void run() {                                                                                 // :1058-1161
        size_t idx = 0;
        for (const Package *pkg : cs.packages)
            for (const BuiltinModule &b : pkg->builtins) {
                ModuleResult mr;
                mr.num_outputs = b.num_outputs; mr.num_temps = b.num_temps; mr.builtin = true;
                cs.module_results[idx] = mr;
                module_done[idx] = true;
                idx++;
            }
        for (size_t gi = 0; gi < globals.size(); gi++) {
            if (global_visited[gi]) continue;
            global_visited[gi] = true;
            global_results[gi] = {true, gen_expression(nullptr, globals[gi].value, nullptr)};
        }
        for (size_t gi = 0; gi < globals.size(); gi++) {
            const Res &r = global_results[gi].second;
            if (r.kind == RK::literal_module && !modules[r.index].builtin) cs.exported_modules.push_back({globals[gi].name, r.index});
        }
    }
};

}  // namespace

void codegen(CompiledScript &cs) {
    CodeGen g(cs);
    g.run();
}

std::unique_ptr<CompiledScript> compile(const std::string &contents, const std::string &filename, const std::vector<const Package *> &packages) {
    std::unique_ptr<CompiledScript> cs(new CompiledScript());
    cs->source = Source{filename, contents};
    cs->packages = packages;
    parse(cs->source, packages, cs->pr);
    codegen(*cs);
    return cs;
}

}  // namespace zs
