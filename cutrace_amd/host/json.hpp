// json.hpp — small self-contained JSON reader for the scene loader.
//
// Replaces the reference's third-party picojson dependency (CMakeLists.txt:14-17,
// not available offline).  Only what the loader needs: null/bool/number/string/
// array/object, numbers held as double (picojson's default without
// PICOJSON_USE_INT64), objects as ordered maps where a repeated key overwrites
// the earlier one (std::map semantics picojson::object has).
#ifndef CUTRACE_AMD_JSON_HPP
#define CUTRACE_AMD_JSON_HPP

#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace cutrace::json {

struct value;
using array = std::vector<value>;
using object = std::map<std::string, value>;

struct value {
  enum kind_t { null_k, bool_k, number_k, string_k, array_k, object_k } kind = null_k;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::shared_ptr<array> arr;
  std::shared_ptr<object> obj;

  bool is_number() const { return kind == number_k; }
  bool is_string() const { return kind == string_k; }
  bool is_array() const { return kind == array_k; }
  bool is_object() const { return kind == object_k; }
};

class parser {
public:
  explicit parser(const std::string &text) : s(text) {}

  // returns true on success; on failure `error` holds a message
  bool parse(value &out) {
    skip_ws();
    if (!parse_value(out, 0)) return false;
    skip_ws();
    if (pos != s.size()) return fail("trailing characters after JSON value");
    return true;
  }
  std::string error;

private:
  const std::string &s;
  size_t pos = 0;

  bool fail(const std::string &msg) {
    if (error.empty()) {
      size_t line = 1;
      for (size_t i = 0; i < pos && i < s.size(); i++)
        if (s[i] == '\n') line++;
      error = "syntax error at line " + std::to_string(line) + ": " + msg;
    }
    return false;
  }
  void skip_ws() {
    while (pos < s.size() && (s[pos] == ' ' || s[pos] == '\t' || s[pos] == '\n' || s[pos] == '\r')) pos++;
  }
  bool literal(const char *lit) {
    size_t n = 0;
    while (lit[n]) n++;
    if (s.compare(pos, n, lit) == 0) { pos += n; return true; }
    return false;
  }
  static void append_utf8(std::string &o, unsigned cp) {
    if (cp < 0x80) o += (char)cp;
    else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
    else if (cp < 0x10000) { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
    else { o += (char)(0xF0 | (cp >> 18)); o += (char)(0x80 | ((cp >> 12) & 0x3F)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
  }
  bool parse_hex4(unsigned &cp) {
    if (pos + 4 > s.size()) return fail("bad \\u escape");
    cp = 0;
    for (int i = 0; i < 4; i++) {
      char c = s[pos++];
      cp <<= 4;
      if (c >= '0' && c <= '9') cp |= (unsigned)(c - '0');
      else if (c >= 'a' && c <= 'f') cp |= (unsigned)(c - 'a' + 10);
      else if (c >= 'A' && c <= 'F') cp |= (unsigned)(c - 'A' + 10);
      else return fail("bad \\u escape");
    }
    return true;
  }
  bool parse_string(std::string &out) {
    if (pos >= s.size() || s[pos] != '"') return fail("expected '\"'");
    pos++;
    out.clear();
    while (pos < s.size()) {
      char c = s[pos++];
      if (c == '"') return true;
      if ((unsigned char)c < 0x20) return fail("control character in string");
      if (c != '\\') { out += c; continue; }
      if (pos >= s.size()) break;
      char e = s[pos++];
      switch (e) {
        case '"': out += '"'; break;
        case '\\': out += '\\'; break;
        case '/': out += '/'; break;
        case 'b': out += '\b'; break;
        case 'f': out += '\f'; break;
        case 'n': out += '\n'; break;
        case 'r': out += '\r'; break;
        case 't': out += '\t'; break;
        case 'u': {
          unsigned cp;
          if (!parse_hex4(cp)) return false;
          if (cp >= 0xD800 && cp <= 0xDBFF) {
            unsigned lo;
            if (!(pos + 1 < s.size() && s[pos] == '\\' && s[pos + 1] == 'u')) return fail("unpaired surrogate");
            pos += 2;
            if (!parse_hex4(lo)) return false;
            if (lo < 0xDC00 || lo > 0xDFFF) return fail("unpaired surrogate");
            cp = 0x10000 + (((cp - 0xD800) << 10) | (lo - 0xDC00));
          }
          append_utf8(out, cp);
          break;
        }
        default: return fail("bad escape in string");
      }
    }
    return fail("unterminated string");
  }
  bool parse_number(value &out) {
    size_t start = pos;
    if (pos < s.size() && s[pos] == '-') pos++;
    if (pos >= s.size() || !(s[pos] >= '0' && s[pos] <= '9')) return fail("bad number");
    while (pos < s.size() && s[pos] >= '0' && s[pos] <= '9') pos++;
    if (pos < s.size() && s[pos] == '.') {
      pos++;
      if (pos >= s.size() || !(s[pos] >= '0' && s[pos] <= '9')) return fail("bad number");
      while (pos < s.size() && s[pos] >= '0' && s[pos] <= '9') pos++;
    }
    if (pos < s.size() && (s[pos] == 'e' || s[pos] == 'E')) {
      pos++;
      if (pos < s.size() && (s[pos] == '+' || s[pos] == '-')) pos++;
      if (pos >= s.size() || !(s[pos] >= '0' && s[pos] <= '9')) return fail("bad number");
      while (pos < s.size() && s[pos] >= '0' && s[pos] <= '9') pos++;
    }
    out.kind = value::number_k;
    out.num = std::strtod(s.substr(start, pos - start).c_str(), nullptr);  // correctly rounded, as picojson's strtod
    return true;
  }
  bool parse_value(value &out, int depth) {
    if (depth > 256) return fail("nesting too deep");
    skip_ws();
    if (pos >= s.size()) return fail("unexpected end of input");
    char c = s[pos];
    if (c == '{') {
      pos++;
      out.kind = value::object_k;
      out.obj = std::make_shared<object>();
      skip_ws();
      if (pos < s.size() && s[pos] == '}') { pos++; return true; }
      for (;;) {
        skip_ws();
        std::string key;
        if (!parse_string(key)) return false;
        skip_ws();
        if (pos >= s.size() || s[pos] != ':') return fail("expected ':'");
        pos++;
        value v;
        if (!parse_value(v, depth + 1)) return false;
        (*out.obj)[key] = std::move(v);
        skip_ws();
        if (pos < s.size() && s[pos] == ',') { pos++; continue; }
        if (pos < s.size() && s[pos] == '}') { pos++; return true; }
        return fail("expected ',' or '}'");
      }
    }
    if (c == '[') {
      pos++;
      out.kind = value::array_k;
      out.arr = std::make_shared<array>();
      skip_ws();
      if (pos < s.size() && s[pos] == ']') { pos++; return true; }
      for (;;) {
        value v;
        if (!parse_value(v, depth + 1)) return false;
        out.arr->push_back(std::move(v));
        skip_ws();
        if (pos < s.size() && s[pos] == ',') { pos++; continue; }
        if (pos < s.size() && s[pos] == ']') { pos++; return true; }
        return fail("expected ',' or ']'");
      }
    }
    if (c == '"') { out.kind = value::string_k; return parse_string(out.str); }
    if (literal("true")) { out.kind = value::bool_k; out.b = true; return true; }
    if (literal("false")) { out.kind = value::bool_k; out.b = false; return true; }
    if (literal("null")) { out.kind = value::null_k; return true; }
    return parse_number(out);
  }
};

}  // namespace cutrace::json
#endif
