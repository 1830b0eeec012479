// Minimal JSON5 reader for scene files.
//
// Replaces the `json5` 0.3.0 crate used at reference src/config.rs:22.  Supports the
// JSON5 features the example scenes rely on (and the rest of the JSON5 grammar that
// is cheap to support): `//` and `/* */` comments, unquoted identifier keys, single-
// and double-quoted strings with escapes, trailing commas, leading `+`/`-`, leading or
// trailing decimal points, hexadecimal integers, Infinity and NaN.
//
// Objects keep their members in file order (the reference deserialises into HashMaps,
// whose order is unspecified; a deterministic order is a refinement, not a change).
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace rt {

struct Json5Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

class JValue {
public:
    enum Kind { Null, Bool, Number, String, Array, Object };
    Kind kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<JValue> arr;
    std::vector<std::pair<std::string, JValue>> obj;
    int line = 0;  // 1-based line where the value starts (for error messages)

    bool is(Kind k) const { return kind == k; }
    const JValue* find(const std::string& key) const {
        for (auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    static const char* kind_name(Kind k);
};

// Parses a whole JSON5 document; throws Json5Error with a line/column message.
JValue json5_parse(const std::string& text);

}  // namespace rt
