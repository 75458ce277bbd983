// zscript.hpp -- zangscript front-end (tokenizer, parser, codegen) and its two backends in C++: the
// compiled-language restatement of src/zangscript/{tokenize,parse,codegen,codegen_zig}.zig that backs the
// C ABI's zh_zscript_* entry points (no device code here; host side of libzang_hip.so).
// zang_amd/zangscript/*.py is the same front-end in Python, kept as an independent second implementation:
// tests/test_zangscript_native.py requires both to print identical Zig and HIP text.
#pragma once
#include <stdint.h>

#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

namespace zs {

// ---- source, locations, errors (context.zig:4-25, fail.zig:47-116)
struct Loc { uint32_t line = 0, index = 0; };
struct SourceRange { Loc loc0, loc1; };
struct Source {
    std::string filename, contents;
    std::string text(const SourceRange &sr) const { return contents.substr(sr.loc0.index, sr.loc1.index - sr.loc0.index); }
};
struct ScriptError {
    std::string rendered;                 // "file:line:col: message" + echoed line + carets
};
[[noreturn]] void fail(const Source &src, const SourceRange &sr, const std::string &message);

// ---- tokens (tokenize.zig:7-33)
enum class TT {
    illegal, end_of_file, name, number, enum_value,
    sym_asterisk, sym_colon, sym_comma, sym_equals, sym_left_paren, sym_minus, sym_plus, sym_right_paren, sym_slash,
    kw_begin, kw_defcurve, kw_defmodule, kw_deftrack, kw_delay, kw_end, kw_false, kw_feedback, kw_from, kw_out, kw_true
};
struct Token { TT tt = TT::illegal; SourceRange sr; float number = 0.0f; };

// ---- builtins (builtins.zig)
struct EnumValue { std::string label; bool f32_payload; };
struct BuiltinEnum { std::string name, zig_name; std::vector<EnumValue> values; };
enum class PK { boolean, buffer, constant, constant_or_buffer, curve, one_of };
struct ParamType { PK kind = PK::constant; const BuiltinEnum *en = nullptr; };
struct ModuleParam { std::string name; ParamType type; };
struct BuiltinModule { std::string name; std::vector<ModuleParam> params; uint32_t num_temps = 0, num_outputs = 1; };
struct Package { std::string zig_package_name, zig_import_path; std::vector<BuiltinModule> builtins; std::vector<const BuiltinEnum *> enums; };
const Package &zang_builtin_package();
const Package &modules_builtin_package();

// ---- parse tree (parse.zig:14-170)
struct NumberLiteral { float value = 0.0f; std::string verbatim; };
struct Expr;
struct Scope;
typedef std::shared_ptr<Expr> ExprP;
struct CallArg { std::string param_name; Token token; ExprP value; };
enum class EK { call, track_call, delay, literal_boolean, literal_number, literal_enum_value, literal_curve, literal_track,
                literal_module, un_arith, bin_arith, local, feedback, name };
struct Expr {
    EK kind;
    SourceRange sr;
    ExprP a, b;                            // operands / call target / track + speed / enum payload
    std::string op;                        // abs cos neg sin sqrt | add div max min mul pow sub
    bool bval = false;
    NumberLiteral num;
    std::string label;                     // enum label
    size_t index = 0;                      // curve / track / module / local index, delay samples
    std::vector<CallArg> args;
    Scope *scope = nullptr;
    Token token;                           // name
};
enum class SK { let_assignment, output, feedback };
struct Statement { SK kind; ExprP expr; size_t local_index = 0; };
struct Scope { Scope *parent = nullptr; std::vector<Statement> statements; };
struct Curve { std::vector<std::pair<NumberLiteral, NumberLiteral>> points; };        // (t, value)
struct TrackNote { NumberLiteral t; SourceRange args_sr; std::vector<CallArg> args; };
struct Track { std::vector<ModuleParam> params; std::vector<TrackNote> notes; };
struct Module {
    std::vector<ModuleParam> params;
    bool builtin = false;
    std::string builtin_name, zig_package_name;
    Scope *scope = nullptr;
    std::vector<std::string> locals;
};
struct Global { std::string name; ExprP value; };
struct ParseResult {
    std::vector<Global> globals;
    std::vector<Curve> curves;
    std::vector<Track> tracks;
    std::vector<Module> modules;
    std::vector<std::unique_ptr<Scope>> scopes;      // owner of every Scope
};
void parse(const Source &src, const std::vector<const Package *> &packages, ParseResult &out);

// ---- codegen (codegen.zig)
struct Res;
typedef std::shared_ptr<Res> ResP;
enum class RK { nothing, temp_buffer, temp_float, literal_boolean, literal_number, literal_enum_value, literal_curve, literal_track,
                literal_module, self_param, track_param };
struct Res {
    RK kind = RK::nothing;
    size_t index = 0;
    bool weak = false;
    bool bval = false;
    NumberLiteral num;
    std::string label;
    ResP payload;
    size_t track_index = 0;
};
struct Dest { bool output = false; size_t index = 0; };
enum class IK { copy_buffer, float_to_buffer, cob_to_buffer, arith_float, arith_buffer, arith_float_float, arith_float_buffer,
                arith_buffer_float, arith_buffer_buffer, call, track_call, delay };
struct Instr {
    IK kind;
    Dest out;                              // buffer destination
    size_t out_float = 0;                  // temp float index (arith_float, arith_float_float)
    std::string op;
    Res a, b, src, speed;
    size_t in_self_param = 0, field_index = 0, track_index = 0, trigger_index = 0, note_tracker_index = 0, delay_index = 0;
    size_t feedback_out_temp = 0, feedback_temp = 0;
    std::vector<size_t> temps;
    std::vector<Res> args;
    std::vector<Instr> instructions;
};
struct ModuleResult {
    size_t num_outputs = 1, num_temps = 0, num_temp_floats = 0;
    bool builtin = false;
    std::vector<size_t> fields, delays, note_trackers, triggers;
    std::vector<Instr> instructions;
};
struct CompiledScript {
    Source source;
    std::vector<const Package *> packages;
    ParseResult pr;
    std::vector<std::vector<std::vector<Res>>> track_results;          // [track][note][param]
    std::vector<ModuleResult> module_results;
    std::vector<std::pair<std::string, size_t>> exported_modules;
};
void codegen(CompiledScript &cs);                                       // fills track_results, module_results, exported_modules
std::unique_ptr<CompiledScript> compile(const std::string &contents, const std::string &filename, const std::vector<const Package *> &packages);

// ---- backends
std::string generate_zig(const CompiledScript &cs);
struct HipParam { std::string name, kind, enum_name; };
struct HipModuleMeta { std::string name, error; size_t state_words = 0, noise_fields = 0, num_temps = 0; std::vector<HipParam> params; };
std::string generate_hip(const CompiledScript &cs, const std::set<std::string> *only, std::vector<HipModuleMeta> &meta, int unroll_override, unsigned forms = 0);

}  // namespace zs
