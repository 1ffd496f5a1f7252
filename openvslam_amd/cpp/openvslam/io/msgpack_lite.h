// msgpack_lite.h -- a reader for the subset of MessagePack that nlohmann::json::to_msgpack emits (maps, arrays, strings, nil, booleans,
// integers, float32 / float64; bin and ext are skipped as nil). Upstream parses map files with nlohmann::json, which this repository
// does not vendor; in an OpenVSLAM checkout io::map_database_io keeps using upstream's own code and this header is not needed.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace openvslam {
namespace io {
namespace msgpack_lite {

struct value {
    enum kind_t { nil, boolean, integer, real, string, array, object } kind = nil;
    bool b = false;
    int64_t i = 0;     // integers (uint64 values above INT64_MAX are not produced by the map format)
    double d = 0.0;
    std::string s;
    std::vector<value> a;
    std::vector<std::pair<std::string, value>> o;   // insertion order kept

    const value& at(const std::string& key) const {
        for (const auto& kv : o)
            if (kv.first == key) return kv.second;
        throw std::runtime_error("msgpack: key '" + key + "' missing");
    }
    bool has(const std::string& key) const {
        for (const auto& kv : o)
            if (kv.first == key) return true;
        return false;
    }
    double num() const {
        if (kind == integer) return (double)i;
        if (kind == real) return d;
        throw std::runtime_error("msgpack: number expected");
    }
    int64_t integer_value() const {
        if (kind == integer) return i;
        if (kind == real && d == (double)(int64_t)d) return (int64_t)d;
        throw std::runtime_error("msgpack: integer expected");
    }
};

class reader {
public:
    reader(const uint8_t* p, size_t n) : p_(p), end_(p + n) {}
    value parse() {
        value v = next(0);
        return v;
    }

private:
    const uint8_t* p_;
    const uint8_t* end_;
    uint8_t u8() {
        if (p_ >= end_) throw std::runtime_error("msgpack: truncated");
        return *p_++;
    }
    uint64_t be(int bytes) {
        if (end_ - p_ < bytes) throw std::runtime_error("msgpack: truncated");
        uint64_t v = 0;
        for (int k = 0; k < bytes; ++k) v = (v << 8) | *p_++;
        return v;
    }
    std::string str(size_t n) {
        if ((size_t)(end_ - p_) < n) throw std::runtime_error("msgpack: truncated");
        std::string s(reinterpret_cast<const char*>(p_), n);
        p_ += n;
        return s;
    }
    value arr(size_t n, int depth) {
        value v;
        v.kind = value::array;
        // n comes straight from the file (up to 2^32 - 1): every element takes at least one byte, so a count beyond the bytes left is a
        // truncated or corrupt file -- say so before reserving for it
        if (n > (size_t)(end_ - p_)) throw std::runtime_error("msgpack: truncated");
        v.a.reserve(n);
        for (size_t k = 0; k < n; ++k) v.a.push_back(next(depth + 1));
        return v;
    }
    value obj(size_t n, int depth) {
        value v;
        v.kind = value::object;
        if (n > (size_t)(end_ - p_) / 2) throw std::runtime_error("msgpack: truncated");   // a key and a value per entry
        v.o.reserve(n);
        for (size_t k = 0; k < n; ++k) {
            value key = next(depth + 1);
            std::string ks = key.kind == value::string ? key.s : (key.kind == value::integer ? std::to_string(key.i) : std::string());
            v.o.emplace_back(std::move(ks), next(depth + 1));
        }
        return v;
    }
    value next(int depth) {
        if (depth > 64) throw std::runtime_error("msgpack: nesting too deep");
        const uint8_t t = u8();
        value v;
        if (t <= 0x7f) { v.kind = value::integer; v.i = t; return v; }
        if (t >= 0xe0) { v.kind = value::integer; v.i = (int8_t)t; return v; }
        if ((t & 0xf0) == 0x80) return obj(t & 0x0f, depth);
        if ((t & 0xf0) == 0x90) return arr(t & 0x0f, depth);
        if ((t & 0xe0) == 0xa0) { v.kind = value::string; v.s = str(t & 0x1f); return v; }
        switch (t) {
            case 0xc0: return v;
            case 0xc2: v.kind = value::boolean; v.b = false; return v;
            case 0xc3: v.kind = value::boolean; v.b = true; return v;
            case 0xc4: str((size_t)be(1)); return v;   // bin: not used by the map format
            case 0xc5: str((size_t)be(2)); return v;
            case 0xc6: str((size_t)be(4)); return v;
            case 0xca: { const uint32_t u = (uint32_t)be(4); float f; std::memcpy(&f, &u, 4); v.kind = value::real; v.d = f; return v; }
            case 0xcb: { const uint64_t u = be(8); std::memcpy(&v.d, &u, 8); v.kind = value::real; return v; }
            case 0xcc: v.kind = value::integer; v.i = (int64_t)be(1); return v;
            case 0xcd: v.kind = value::integer; v.i = (int64_t)be(2); return v;
            case 0xce: v.kind = value::integer; v.i = (int64_t)be(4); return v;
            case 0xcf: v.kind = value::integer; v.i = (int64_t)be(8); return v;
            case 0xd0: v.kind = value::integer; v.i = (int8_t)be(1); return v;
            case 0xd1: v.kind = value::integer; v.i = (int16_t)be(2); return v;
            case 0xd2: v.kind = value::integer; v.i = (int32_t)be(4); return v;
            case 0xd3: v.kind = value::integer; v.i = (int64_t)be(8); return v;
            case 0xd9: v.kind = value::string; v.s = str((size_t)be(1)); return v;
            case 0xda: v.kind = value::string; v.s = str((size_t)be(2)); return v;
            case 0xdb: v.kind = value::string; v.s = str((size_t)be(4)); return v;
            case 0xdc: return arr((size_t)be(2), depth);
            case 0xdd: return arr((size_t)be(4), depth);
            case 0xde: return obj((size_t)be(2), depth);
            case 0xdf: return obj((size_t)be(4), depth);
            default: throw std::runtime_error("msgpack: unsupported type byte");
        }
    }
};

}   // namespace msgpack_lite
}   // namespace io
}   // namespace openvslam
