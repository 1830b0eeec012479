// JSON5 reader (see json5.hpp).  Boundary code: replaces json5::from_str at
// reference src/config.rs:22.
#include "json5.hpp"

#include <cctype>
#include <cmath>
#include <cstdlib>
#include <limits>

namespace rt {

const char* JValue::kind_name(Kind k) {
    switch (k) {
        case Null: return "null";
        case Bool: return "boolean";
        case Number: return "number";
        case String: return "string";
        case Array: return "array";
        case Object: return "object";
    }
    return "?";
}

namespace {

struct Parser {
    const std::string& s;
    size_t i = 0;
    int line = 1, col = 1;

    explicit Parser(const std::string& text) : s(text) {}

    [[noreturn]] void fail(const std::string& msg) const {
        throw Json5Error("JSON5 parse error at line " + std::to_string(line) + ", column " +
                         std::to_string(col) + ": " + msg);
    }
    bool eof() const { return i >= s.size(); }
    char peek(size_t k = 0) const { return i + k < s.size() ? s[i + k] : '\0'; }
    char next() {
        char c = s[i++];
        if (c == '\n') {
            line++;
            col = 1;
        } else {
            col++;
        }
        return c;
    }
    void skip_ws() {
        for (;;) {
            if (eof()) return;
            unsigned char c = (unsigned char)peek();
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f') {
                next();
            } else if (c == 0xC2 && (unsigned char)peek(1) == 0xA0) {  // NBSP
                next();
                next();
            } else if (c == 0xEF && (unsigned char)peek(1) == 0xBB && (unsigned char)peek(2) == 0xBF) {
                next(); next(); next();  // BOM
            } else if (c == '/' && peek(1) == '/') {
                while (!eof() && peek() != '\n') next();
            } else if (c == '/' && peek(1) == '*') {
                next();
                next();
                for (;;) {
                    if (eof()) fail("unterminated block comment");
                    if (peek() == '*' && peek(1) == '/') {
                        next();
                        next();
                        break;
                    }
                    next();
                }
            } else {
                return;
            }
        }
    }

    static bool ident_start(char c) {
        return std::isalpha((unsigned char)c) || c == '_' || c == '$' || ((unsigned char)c >= 0x80);
    }
    static bool ident_part(char c) { return ident_start(c) || std::isdigit((unsigned char)c); }

    static void append_utf8(std::string& out, uint32_t cp) {
        if (cp < 0x80) {
            out += (char)cp;
        } else if (cp < 0x800) {
            out += (char)(0xC0 | (cp >> 6));
            out += (char)(0x80 | (cp & 0x3F));
        } else if (cp < 0x10000) {
            out += (char)(0xE0 | (cp >> 12));
            out += (char)(0x80 | ((cp >> 6) & 0x3F));
            out += (char)(0x80 | (cp & 0x3F));
        } else {
            out += (char)(0xF0 | (cp >> 18));
            out += (char)(0x80 | ((cp >> 12) & 0x3F));
            out += (char)(0x80 | ((cp >> 6) & 0x3F));
            out += (char)(0x80 | (cp & 0x3F));
        }
    }

    uint32_t hex_digits(int n) {
        uint32_t v = 0;
        for (int k = 0; k < n; k++) {
            if (eof()) fail("unterminated escape");
            char c = next();
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else fail("bad hex digit in escape");
        }
        return v;
    }

    std::string parse_string() {
        char quote = next();
        std::string out;
        for (;;) {
            if (eof()) fail("unterminated string");
            char c = next();
            if (c == quote) return out;
            if (c == '\n') fail("newline in string");
            if (c != '\\') {
                out += c;
                continue;
            }
            if (eof()) fail("unterminated string");
            char e = next();
            switch (e) {
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'n': out += '\n'; break;
                case 'r': out += '\r'; break;
                case 't': out += '\t'; break;
                case 'v': out += '\v'; break;
                case '0': out += '\0'; break;
                case 'x': append_utf8(out, hex_digits(2)); break;
                case 'u': {
                    uint32_t cp = hex_digits(4);
                    if (cp >= 0xD800 && cp < 0xDC00 && peek() == '\\' && peek(1) == 'u') {
                        next();
                        next();
                        uint32_t lo = hex_digits(4);
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    append_utf8(out, cp);
                    break;
                }
                case '\r':
                    if (peek() == '\n') next();
                    break;  // line continuation
                case '\n': break;
                default: out += e; break;
            }
        }
    }

    JValue parse_number() {
        JValue v;
        v.kind = JValue::Number;
        v.line = line;
        double sign = 1.0;
        if (peek() == '+' || peek() == '-') {
            if (next() == '-') sign = -1.0;
        }
        if (s.compare(i, 8, "Infinity") == 0) {
            for (int k = 0; k < 8; k++) next();
            v.num = sign * std::numeric_limits<double>::infinity();
            return v;
        }
        if (s.compare(i, 3, "NaN") == 0) {
            for (int k = 0; k < 3; k++) next();
            v.num = std::numeric_limits<double>::quiet_NaN();
            return v;
        }
        if (peek() == '0' && (peek(1) == 'x' || peek(1) == 'X')) {
            next();
            next();
            if (!std::isxdigit((unsigned char)peek())) fail("bad hexadecimal number");
            double acc = 0;
            while (std::isxdigit((unsigned char)peek())) {
                char c = next();
                int d = c <= '9' ? c - '0' : (c | 0x20) - 'a' + 10;
                acc = acc * 16 + d;
            }
            v.num = sign * acc;
            return v;
        }
        size_t start = i;
        bool digits = false;
        while (std::isdigit((unsigned char)peek())) {
            next();
            digits = true;
        }
        if (peek() == '.') {
            next();
            while (std::isdigit((unsigned char)peek())) {
                next();
                digits = true;
            }
        }
        if (!digits) fail("expected a number");
        if (peek() == 'e' || peek() == 'E') {
            next();
            if (peek() == '+' || peek() == '-') next();
            if (!std::isdigit((unsigned char)peek())) fail("bad exponent");
            while (std::isdigit((unsigned char)peek())) next();
        }
        std::string tok = s.substr(start, i - start);
        if (tok[0] == '.') tok = "0" + tok;
        v.num = sign * std::strtod(tok.c_str(), nullptr);
        return v;
    }

    JValue parse_value(int depth) {
        if (depth > 256) fail("nesting too deep");
        skip_ws();
        if (eof()) fail("unexpected end of input");
        char c = peek();
        JValue v;
        v.line = line;
        if (c == '{') {
            next();
            v.kind = JValue::Object;
            for (;;) {
                skip_ws();
                if (eof()) fail("unterminated object");
                if (peek() == '}') {
                    next();
                    return v;
                }
                std::string key;
                if (peek() == '"' || peek() == '\'') {
                    key = parse_string();
                } else if (ident_start(peek())) {
                    while (!eof() && ident_part(peek())) key += next();
                } else {
                    fail(std::string("expected an object key, found '") + peek() + "'");
                }
                skip_ws();
                if (eof() || next() != ':') fail("expected ':' after object key");
                JValue member = parse_value(depth + 1);
                bool replaced = false;
                for (auto& kv : v.obj) {
                    if (kv.first == key) {  // serde map semantics: the last duplicate wins
                        kv.second = std::move(member);
                        replaced = true;
                        break;
                    }
                }
                if (!replaced) v.obj.emplace_back(std::move(key), std::move(member));
                skip_ws();
                if (peek() == ',') {
                    next();
                } else if (peek() != '}') {
                    fail("expected ',' or '}' in object");
                }
            }
        }
        if (c == '[') {
            next();
            v.kind = JValue::Array;
            for (;;) {
                skip_ws();
                if (eof()) fail("unterminated array");
                if (peek() == ']') {
                    next();
                    return v;
                }
                v.arr.push_back(parse_value(depth + 1));
                skip_ws();
                if (peek() == ',') {
                    next();
                } else if (peek() != ']') {
                    fail("expected ',' or ']' in array");
                }
            }
        }
        if (c == '"' || c == '\'') {
            v.kind = JValue::String;
            v.str = parse_string();
            return v;
        }
        if (std::isdigit((unsigned char)c) || c == '.' || c == '+' || c == '-' || c == 'I' || c == 'N') {
            return parse_number();
        }
        if (s.compare(i, 4, "true") == 0 && !ident_part(peek(4))) {
            for (int k = 0; k < 4; k++) next();
            v.kind = JValue::Bool;
            v.b = true;
            return v;
        }
        if (s.compare(i, 5, "false") == 0 && !ident_part(peek(5))) {
            for (int k = 0; k < 5; k++) next();
            v.kind = JValue::Bool;
            v.b = false;
            return v;
        }
        if (s.compare(i, 4, "null") == 0 && !ident_part(peek(4))) {
            for (int k = 0; k < 4; k++) next();
            v.kind = JValue::Null;
            return v;
        }
        fail(std::string("unexpected character '") + c + "'");
    }
};

}  // namespace

JValue json5_parse(const std::string& text) {
    Parser p(text);
    JValue v = p.parse_value(0);
    p.skip_ws();
    if (!p.eof()) p.fail("trailing characters after the document");
    return v;
}

}  // namespace rt
