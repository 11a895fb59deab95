// Minimal JSON DOM parser (RFC 8259 subset sufficient for render_option.json and glTF 2.0).
// Replaces the reference's un-vendored nlohmann/json + tinygltf JSON layer (.gitmodules:1-12).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace hjr {

struct JsonError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

class Json {
public:
    enum Type { Null, Bool, Number, String, Array, Object };
    Type type = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj; // insertion order kept (glTF extension iteration order)

    bool is_null() const { return type == Null; }
    bool is_object() const { return type == Object; }
    bool is_array() const { return type == Array; }
    bool is_number() const { return type == Number; }
    bool is_string() const { return type == String; }
    bool is_bool() const { return type == Bool; }
    size_t size() const { return type == Array ? arr.size() : (type == Object ? obj.size() : 0); }

    const Json* find(const std::string& k) const
    {
        if (type != Object) return nullptr;
        for (auto& kv : obj)
            if (kv.first == k) return &kv.second;
        return nullptr;
    }
    bool has(const std::string& k) const { return find(k) != nullptr; }
    // nlohmann-like access: a missing key is an error (render_json_loader.h:222-225 turns it into `false`)
    const Json& at(const std::string& k) const
    {
        const Json* j = find(k);
        if (!j) throw JsonError("missing key '" + k + "'");
        return *j;
    }
    const Json& at(size_t i) const
    {
        if (type != Array || i >= arr.size()) throw JsonError("array index out of range");
        return arr[i];
    }
    double as_number() const
    {
        if (type != Number) throw JsonError("expected a number");
        return num;
    }
    int64_t as_int() const { return (int64_t)as_number(); }
    bool as_bool() const
    {
        if (type != Bool) throw JsonError("expected a boolean");
        return b;
    }
    const std::string& as_string() const
    {
        if (type != String) throw JsonError("expected a string");
        return str;
    }
    double number_or(const std::string& k, double d) const
    {
        const Json* j = find(k);
        return (j && j->is_number()) ? j->num : d;
    }
    int64_t int_or(const std::string& k, int64_t d) const
    {
        const Json* j = find(k);
        return (j && j->is_number()) ? (int64_t)j->num : d;
    }
    std::string string_or(const std::string& k, const std::string& d) const
    {
        const Json* j = find(k);
        return (j && j->is_string()) ? j->str : d;
    }

    static Json parse(const std::string& text)
    {
        Parser p{ text, 0 };
        p.skip_ws();
        Json j = p.value(0);
        p.skip_ws();
        if (p.pos != text.size()) throw JsonError("trailing characters after JSON value at offset " + std::to_string(p.pos));
        return j;
    }

private:
    struct Parser {
        const std::string& s;
        size_t pos;
        void skip_ws()
        {
            while (pos < s.size() && (s[pos] == ' ' || s[pos] == '\t' || s[pos] == '\n' || s[pos] == '\r')) pos++;
        }
        [[noreturn]] void fail(const std::string& m) { throw JsonError(m + " at offset " + std::to_string(pos)); }
        Json value(int depth)
        {
            if (depth > 200) fail("nesting too deep");
            if (pos >= s.size()) fail("unexpected end of input");
            char c = s[pos];
            if (c == '{') return object(depth);
            if (c == '[') return array(depth);
            if (c == '"') {
                Json j;
                j.type = String;
                j.str = string();
                return j;
            }
            if (c == 't' || c == 'f' || c == 'n') return literal();
            return number();
        }
        Json literal()
        {
            Json j;
            if (s.compare(pos, 4, "true") == 0) { j.type = Bool; j.b = true; pos += 4; }
            else if (s.compare(pos, 5, "false") == 0) { j.type = Bool; j.b = false; pos += 5; }
            else if (s.compare(pos, 4, "null") == 0) { j.type = Null; pos += 4; }
            else fail("invalid literal");
            return j;
        }
        Json number()
        {
            size_t st = pos;
            if (pos < s.size() && (s[pos] == '-' || s[pos] == '+')) pos++;
            bool digits = false;
            while (pos < s.size() && ((s[pos] >= '0' && s[pos] <= '9') || s[pos] == '.' || s[pos] == 'e' || s[pos] == 'E' ||
                                      s[pos] == '-' || s[pos] == '+')) {
                if (s[pos] >= '0' && s[pos] <= '9') digits = true;
                pos++;
            }
            if (!digits) fail("invalid number");
            Json j;
            j.type = Number;
            j.num = std::strtod(s.substr(st, pos - st).c_str(), nullptr);
            return j;
        }
        static void put_utf8(std::string& o, uint32_t cp)
        {
            if (cp < 0x80) o += (char)cp;
            else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
            else if (cp < 0x10000) { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
            else { o += (char)(0xF0 | (cp >> 18)); o += (char)(0x80 | ((cp >> 12) & 0x3F)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
        }
        uint32_t hex4()
        {
            if (pos + 4 > s.size()) fail("truncated \\u escape");
            uint32_t v = 0;
            for (int i = 0; i < 4; i++) {
                char c = s[pos++];
                v <<= 4;
                if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
                else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
                else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
                else fail("bad hex digit");
            }
            return v;
        }
        std::string string()
        {
            std::string o;
            pos++; // opening quote
            for (;;) {
                if (pos >= s.size()) fail("unterminated string");
                char c = s[pos++];
                if (c == '"') break;
                if (c == '\\') {
                    if (pos >= s.size()) fail("unterminated escape");
                    char e = s[pos++];
                    switch (e) {
                    case '"': o += '"'; break;
                    case '\\': o += '\\'; break;
                    case '/': o += '/'; break;
                    case 'b': o += '\b'; break;
                    case 'f': o += '\f'; break;
                    case 'n': o += '\n'; break;
                    case 'r': o += '\r'; break;
                    case 't': o += '\t'; break;
                    case 'u': {
                        uint32_t cp = hex4();
                        if (cp >= 0xD800 && cp < 0xDC00 && pos + 1 < s.size() && s[pos] == '\\' && s[pos + 1] == 'u') {
                            pos += 2;
                            uint32_t lo = hex4();
                            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        }
                        put_utf8(o, cp);
                        break;
                    }
                    default: fail("bad escape");
                    }
                } else o += c;
            }
            return o;
        }
        Json array(int depth)
        {
            Json j;
            j.type = Array;
            pos++;
            skip_ws();
            if (pos < s.size() && s[pos] == ']') { pos++; return j; }
            for (;;) {
                skip_ws();
                j.arr.push_back(value(depth + 1));
                skip_ws();
                if (pos >= s.size()) fail("unterminated array");
                if (s[pos] == ',') { pos++; continue; }
                if (s[pos] == ']') { pos++; break; }
                fail("expected ',' or ']'");
            }
            return j;
        }
        Json object(int depth)
        {
            Json j;
            j.type = Object;
            pos++;
            skip_ws();
            if (pos < s.size() && s[pos] == '}') { pos++; return j; }
            for (;;) {
                skip_ws();
                if (pos >= s.size() || s[pos] != '"') fail("expected a key string");
                std::string k = string();
                skip_ws();
                if (pos >= s.size() || s[pos] != ':') fail("expected ':'");
                pos++;
                skip_ws();
                j.obj.emplace_back(std::move(k), value(depth + 1));
                skip_ws();
                if (pos >= s.size()) fail("unterminated object");
                if (s[pos] == ',') { pos++; continue; }
                if (s[pos] == '}') { pos++; break; }
                fail("expected ',' or '}'");
            }
            return j;
        }
    };
};

} // namespace hjr
