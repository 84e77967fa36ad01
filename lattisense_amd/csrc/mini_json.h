// mini_json.h — small JSON DOM reader for mega_ag.json / task_signature.json (objects keep insertion order).
#pragma once
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace mjson {

struct Value;
using ValuePtr = std::shared_ptr<Value>;

struct Value {
    enum Kind { Null, Bool, Int, Float, String, Array, Object } kind = Null;
    bool b = false;
    int64_t i = 0;        // also holds uint64 bit patterns (is_unsigned)
    bool is_unsigned = false;
    double f = 0.0;
    std::string s;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;

    bool is_null() const { return kind == Null; }
    bool contains(const std::string& k) const {
        if (kind != Object) return false;
        for (auto& kv : obj)
            if (kv.first == k) return true;
        return false;
    }
    const Value& at(const std::string& k) const {
        if (kind != Object) throw std::runtime_error("json: not an object (key " + k + ")");
        for (auto& kv : obj)
            if (kv.first == k) return kv.second;
        throw std::runtime_error("json: missing key " + k);
    }
    const Value& operator[](const std::string& k) const { return at(k); }
    const Value& operator[](size_t idx) const {
        if (kind != Array || idx >= arr.size()) throw std::runtime_error("json: bad array index");
        return arr[idx];
    }
    size_t size() const { return kind == Array ? arr.size() : kind == Object ? obj.size() : 0; }
    int64_t as_int() const {
        if (kind == Int) return i;
        if (kind == Float) return (int64_t)f;
        throw std::runtime_error("json: not a number");
    }
    uint64_t as_u64() const {
        if (kind == Int) return (uint64_t)i;
        if (kind == Float) return (uint64_t)f;
        throw std::runtime_error("json: not a number");
    }
    double as_double() const {
        if (kind == Float) return f;
        if (kind == Int) return is_unsigned ? (double)(uint64_t)i : (double)i;
        throw std::runtime_error("json: not a number");
    }
    bool as_bool() const {
        if (kind == Bool) return b;
        throw std::runtime_error("json: not a bool");
    }
    const std::string& as_string() const {
        if (kind == String) return s;
        throw std::runtime_error("json: not a string");
    }
    std::vector<uint64_t> as_u64_vector() const {
        std::vector<uint64_t> v;
        if (kind != Array) throw std::runtime_error("json: not an array");
        for (auto& e : arr) v.push_back(e.as_u64());
        return v;
    }
};

class Parser {
public:
    explicit Parser(const std::string& text) : t(text) {}
    Value parse() {
        Value v = value();
        ws();
        if (p != t.size()) fail("trailing characters");
        return v;
    }

private:
    const std::string& t;
    size_t p = 0;
    int depth = 0;
    static constexpr int kMaxDepth = 256;  // task files nest 5 deep; the bound keeps a hostile file from exhausting the stack
    struct Nest {
        Parser& ps;
        explicit Nest(Parser& q) : ps(q) {
            if (++ps.depth > kMaxDepth) ps.fail("nesting too deep");
        }
        ~Nest() { ps.depth--; }
    };
    [[noreturn]] void fail(const std::string& m) { throw std::runtime_error("json parse error at " + std::to_string(p) + ": " + m); }
    void ws() {
        while (p < t.size() && (t[p] == ' ' || t[p] == '\n' || t[p] == '\t' || t[p] == '\r')) p++;
    }
    Value value() {
        ws();
        if (p >= t.size()) fail("unexpected end");
        char c = t[p];
        if (c == '{') {
            Nest n(*this);
            return object();
        }
        if (c == '[') {
            Nest n(*this);
            return array();
        }
        if (c == '"') {
            Value v;
            v.kind = Value::String;
            v.s = string();
            return v;
        }
        if (t.compare(p, 4, "true") == 0) {
            p += 4;
            Value v;
            v.kind = Value::Bool;
            v.b = true;
            return v;
        }
        if (t.compare(p, 5, "false") == 0) {
            p += 5;
            Value v;
            v.kind = Value::Bool;
            return v;
        }
        if (t.compare(p, 4, "null") == 0) {
            p += 4;
            return Value();
        }
        return number();
    }
    Value number() {
        size_t st = p;
        bool is_float = false;
        if (t[p] == '-') p++;
        while (p < t.size() && (isdigit((unsigned char)t[p]) || t[p] == '.' || t[p] == 'e' || t[p] == 'E' || t[p] == '+' || t[p] == '-')) {
            if (t[p] == '.' || t[p] == 'e' || t[p] == 'E') is_float = true;
            p++;
        }
        if (st == p) fail("bad number");
        std::string tok = t.substr(st, p - st);
        Value v;
        char* end = nullptr;
        errno = 0;
        if (is_float) {
            v.kind = Value::Float;
            v.f = strtod(tok.c_str(), &end);
        } else {
            v.kind = Value::Int;
            if (tok[0] == '-') {
                v.i = strtoll(tok.c_str(), &end, 10);
            } else {
                v.i = (int64_t)strtoull(tok.c_str(), &end, 10);
                v.is_unsigned = true;
            }
        }
        bool overflow = errno == ERANGE && (!is_float || std::isinf(v.f));  // a float that underflows to 0 is fine
        if (end != tok.c_str() + tok.size() || overflow) {
            p = st;
            fail("bad number '" + tok + "'");
        }
        return v;
    }
    std::string string() {
        std::string out;
        p++;  // opening quote
        while (p < t.size() && t[p] != '"') {
            if (t[p] == '\\') {
                p++;
                if (p >= t.size()) fail("bad escape");
                switch (t[p]) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {  // keep BMP code points as UTF-8
                        if (p + 4 >= t.size()) fail("bad unicode escape");
                        unsigned cp = (unsigned)strtoul(t.substr(p + 1, 4).c_str(), nullptr, 16);
                        p += 4;
                        if (cp < 0x80) out += (char)cp;
                        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                        else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                        break;
                    }
                    default: out += t[p];
                }
                p++;
            } else {
                out += t[p++];
            }
        }
        if (p >= t.size()) fail("unterminated string");
        p++;
        return out;
    }
    Value array() {
        Value v;
        v.kind = Value::Array;
        p++;
        ws();
        if (p < t.size() && t[p] == ']') { p++; return v; }
        while (true) {
            v.arr.push_back(value());
            ws();
            if (p >= t.size()) fail("unterminated array");
            if (t[p] == ',') { p++; continue; }
            if (t[p] == ']') { p++; break; }
            fail("expected , or ]");
        }
        return v;
    }
    Value object() {
        Value v;
        v.kind = Value::Object;
        p++;
        ws();
        if (p < t.size() && t[p] == '}') { p++; return v; }
        while (true) {
            ws();
            if (p >= t.size() || t[p] != '"') fail("expected key");
            std::string k = string();
            ws();
            if (p >= t.size() || t[p] != ':') fail("expected :");
            p++;
            v.obj.emplace_back(std::move(k), value());
            ws();
            if (p >= t.size()) fail("unterminated object");
            if (t[p] == ',') { p++; continue; }
            if (t[p] == '}') { p++; break; }
            fail("expected , or }");
        }
        return v;
    }
};

inline Value parse_file(const std::string& path) {
    std::ifstream f(path);
    if (!f.is_open()) throw std::runtime_error("Cannot open " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    std::string text = ss.str();
    return Parser(text).parse();
}

}  // namespace mjson
