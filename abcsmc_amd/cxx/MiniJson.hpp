// MiniJson.hpp -- small JSON reader for the AbcSmc configuration files (the reference links jsoncpp, which is a
// submodule absent from the tree: /root/reference/.gitmodules, src/AbcSmc.cpp:276-291).  Accepts what jsoncpp's
// default Reader accepts for these files: objects, arrays, strings with escapes, numbers, true/false/null and
// C / C++ comments.  Accessors follow the calls AbcSmc.cpp makes on Json::Value (isMember, get, operator[],
// asString/asDouble/asInt64/asUInt64/asBool, array iteration), with jsoncpp's conversion rules for the cases the
// configuration uses (a missing key is a null value; null converts to 0 / "" / false).
#ifndef ABCSMC_AMD_MINIJSON_HPP
#define ABCSMC_AMD_MINIJSON_HPP

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace mjson {

struct ParseError : std::runtime_error {
    size_t line;
    ParseError(const std::string& m, size_t l) : std::runtime_error("line " + std::to_string(l) + ": " + m), line(l) {}
};

class Value {
   public:
    enum Type { Null, Bool, Number, String, Array, Object };
    Value() : t_(Null), b_(false), n_(0) {}
    Value(bool b) : t_(Bool), b_(b), n_(0) {}
    Value(double n) : t_(Number), b_(false), n_(n) {}
    Value(int n) : t_(Number), b_(false), n_(n) {}
    Value(const char* s) : t_(String), b_(false), n_(0), s_(s) {}
    Value(const std::string& s) : t_(String), b_(false), n_(0), s_(s) {}

    Type type() const { return t_; }
    bool isNull() const { return t_ == Null; }
    bool isArray() const { return t_ == Array; }
    bool isObject() const { return t_ == Object; }
    bool isString() const { return t_ == String; }
    bool isNumeric() const { return t_ == Number || t_ == Bool; }

    bool isMember(const std::string& k) const { return t_ == Object && o_.count(k) != 0; }
    const Value& operator[](const std::string& k) const {
        if (t_ == Object) { auto it = o_.find(k); if (it != o_.end()) return it->second; }
        return null_value();
    }
    const Value& operator[](const char* k) const { return (*this)[std::string(k)]; }
    const Value& operator[](size_t i) const { return (t_ == Array && i < a_.size()) ? a_[i] : null_value(); }
    const Value& operator[](int i) const { return (*this)[(size_t)i]; }
    template <typename T> Value get(const std::string& k, const T& dflt) const { return isMember(k) ? (*this)[k] : Value(dflt); }
    size_t size() const { return t_ == Array ? a_.size() : (t_ == Object ? o_.size() : 0); }
    std::vector<Value>::const_iterator begin() const { return a_.begin(); }   // arrays only (objects: members())
    std::vector<Value>::const_iterator end() const { return a_.end(); }
    const std::map<std::string, Value>& members() const { return o_; }

    std::string asString() const {
        switch (t_) {
            case String: return s_;
            case Null: return "";
            case Bool: return b_ ? "true" : "false";
            case Number: { char buf[40]; snprintf(buf, sizeof buf, "%.17g", n_); return buf; }
            default: throw std::runtime_error("json: value is not convertible to string");
        }
    }
    double asDouble() const {
        switch (t_) {
            case Number: return n_;
            case Null: return 0.0;
            case Bool: return b_ ? 1.0 : 0.0;
            default: throw std::runtime_error("json: value is not convertible to double");
        }
    }
    int64_t asInt64() const {
        const double d = asDouble();
        if (d != std::floor(d)) throw std::runtime_error("json: value is not integral");
        return (int64_t)d;
    }
    uint64_t asUInt64() const {
        const double d = asDouble();
        if (d < 0 || d != std::floor(d)) throw std::runtime_error("json: value is not an unsigned integer");
        return (uint64_t)d;
    }
    int asInt() const { return (int)asInt64(); }
    bool asBool() const {
        switch (t_) {
            case Bool: return b_;
            case Null: return false;
            case Number: return n_ != 0.0;
            default: throw std::runtime_error("json: value is not convertible to bool");
        }
    }
    template <typename T> T as() const;

    // construction (parser)
    static Value array() { Value v; v.t_ = Array; return v; }
    static Value object() { Value v; v.t_ = Object; return v; }
    void push(const Value& v) { a_.push_back(v); }
    void set(const std::string& k, const Value& v) { o_[k] = v; }

   private:
    static const Value& null_value() { static const Value v; return v; }
    Type t_;
    bool b_;
    double n_;
    std::string s_;
    std::vector<Value> a_;
    std::map<std::string, Value> o_;
};
template <> inline double Value::as<double>() const { return asDouble(); }
template <> inline float Value::as<float>() const { return (float)asDouble(); }
template <> inline size_t Value::as<size_t>() const { return (size_t)asUInt64(); }
template <> inline int Value::as<int>() const { return asInt(); }
template <> inline long Value::as<long>() const { return (long)asInt64(); }
template <> inline std::string Value::as<std::string>() const { return asString(); }
template <> inline bool Value::as<bool>() const { return asBool(); }

class Parser {
   public:
    explicit Parser(const std::string& text) : s_(text), i_(0), line_(1) {}
    Value parse_document() {
        Value v = value();
        ws();
        if (i_ != s_.size()) fail("trailing characters after the document");
        return v;
    }

   private:
    const std::string& s_;
    size_t i_, line_;
    [[noreturn]] void fail(const std::string& m) const { throw ParseError(m, line_); }
    void ws() {
        for (;;) {
            while (i_ < s_.size() && (s_[i_] == ' ' || s_[i_] == '\t' || s_[i_] == '\r' || s_[i_] == '\n')) {
                if (s_[i_] == '\n') line_++;
                i_++;
            }
            if (i_ + 1 < s_.size() && s_[i_] == '/' && s_[i_ + 1] == '/') {
                while (i_ < s_.size() && s_[i_] != '\n') i_++;
            } else if (i_ + 1 < s_.size() && s_[i_] == '/' && s_[i_ + 1] == '*') {
                i_ += 2;
                while (i_ + 1 < s_.size() && !(s_[i_] == '*' && s_[i_ + 1] == '/')) { if (s_[i_] == '\n') line_++; i_++; }
                if (i_ + 1 >= s_.size()) fail("unterminated comment");
                i_ += 2;
            } else {
                return;
            }
        }
    }
    Value value() {
        ws();
        if (i_ >= s_.size()) fail("unexpected end of input");
        const char c = s_[i_];
        if (c == '{') return object();
        if (c == '[') return array();
        if (c == '"') return Value(string());
        if (s_.compare(i_, 4, "true") == 0) { i_ += 4; return Value(true); }
        if (s_.compare(i_, 5, "false") == 0) { i_ += 5; return Value(false); }
        if (s_.compare(i_, 4, "null") == 0) { i_ += 4; return Value(); }
        if (c == '-' || c == '+' || (c >= '0' && c <= '9') || c == '.') return number();
        fail(std::string("unexpected character '") + c + "'");
    }
    Value number() {
        const char* b = s_.c_str() + i_;
        char* e = nullptr;
        const double d = std::strtod(b, &e);
        if (e == b) fail("malformed number");
        i_ += (size_t)(e - b);
        return Value(d);
    }
    static void utf8(std::string& out, unsigned cp) {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
        else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
    }
    unsigned hex4() {
        if (i_ + 4 > s_.size()) fail("truncated \\u escape");
        unsigned v = 0;
        for (int k = 0; k < 4; k++) {
            const char c = s_[i_++];
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
            else fail("bad hex digit in \\u escape");
        }
        return v;
    }
    std::string string() {
        std::string out;
        i_++;   // opening quote
        for (;;) {
            if (i_ >= s_.size()) fail("unterminated string");
            const char c = s_[i_++];
            if (c == '"') return out;
            if (c == '\n') line_++;
            if (c != '\\') { out += c; continue; }
            if (i_ >= s_.size()) fail("unterminated escape");
            const char e = s_[i_++];
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
                    unsigned cp = hex4();
                    if (cp >= 0xD800 && cp < 0xDC00 && i_ + 1 < s_.size() && s_[i_] == '\\' && s_[i_ + 1] == 'u') {
                        i_ += 2;
                        const unsigned lo = hex4();
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    utf8(out, cp);
                    break;
                }
                default: fail("unknown escape");
            }
        }
    }
    Value array() {
        Value v = Value::array();
        i_++;
        ws();
        if (i_ < s_.size() && s_[i_] == ']') { i_++; return v; }
        for (;;) {
            v.push(value());
            ws();
            if (i_ >= s_.size()) fail("unterminated array");
            if (s_[i_] == ',') { i_++; continue; }
            if (s_[i_] == ']') { i_++; return v; }
            fail("expected ',' or ']'");
        }
    }
    Value object() {
        Value v = Value::object();
        i_++;
        ws();
        if (i_ < s_.size() && s_[i_] == '}') { i_++; return v; }
        for (;;) {
            ws();
            if (i_ >= s_.size() || s_[i_] != '"') fail("expected a member name");
            const std::string k = string();
            ws();
            if (i_ >= s_.size() || s_[i_] != ':') fail("expected ':'");
            i_++;
            v.set(k, value());
            ws();
            if (i_ >= s_.size()) fail("unterminated object");
            if (s_[i_] == ',') { i_++; continue; }
            if (s_[i_] == '}') { i_++; return v; }
            fail("expected ',' or '}'");
        }
    }
};

inline Value parse(const std::string& text) { return Parser(text).parse_document(); }

}  // namespace mjson
#endif
