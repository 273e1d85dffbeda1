// Minimal JSON reader for the two places the reference uses a JSON library: config.json
// (nlohmann::json, src/main.cc:24-61) and the glTF document (tinygltf's embedded parser,
// src/gltf_loader/gltf_loader.cc:121-143). Objects keep insertion order; numbers are doubles.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace ptjson {

class Value {
 public:
  enum class Kind { Null, Bool, Number, String, Array, Object };

  Kind kind = Kind::Null;
  bool boolean = false;
  double number = 0.0;
  std::string string;
  std::vector<Value> array;
  std::vector<std::pair<std::string, Value>> object;

  bool is_null() const { return kind == Kind::Null; }
  bool is_number() const { return kind == Kind::Number; }
  bool is_string() const { return kind == Kind::String; }
  bool is_array() const { return kind == Kind::Array; }
  bool is_object() const { return kind == Kind::Object; }

  const Value* find(const std::string& key) const {
    if (kind != Kind::Object) return nullptr;
    for (const auto& kv : object)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
  bool contains(const std::string& key) const { return find(key) != nullptr; }
  const Value& at(const std::string& key) const {
    const Value* v = find(key);
    if (!v) throw std::runtime_error("JSON: missing key '" + key + "'");
    return *v;
  }
  const Value& at(size_t i) const {
    if (kind != Kind::Array || i >= array.size()) throw std::runtime_error("JSON: array index out of range");
    return array[i];
  }
  size_t size() const { return kind == Kind::Array ? array.size() : (kind == Kind::Object ? object.size() : 0); }

  double as_number() const {
    if (kind != Kind::Number) throw std::runtime_error("JSON: number expected");
    return number;
  }
  int64_t as_int() const { return static_cast<int64_t>(as_number()); }
  const std::string& as_string() const {
    if (kind != Kind::String) throw std::runtime_error("JSON: string expected");
    return string;
  }
  double number_or(const std::string& key, double fallback) const {
    const Value* v = find(key);
    return (v && v->is_number()) ? v->number : fallback;
  }
  int64_t int_or(const std::string& key, int64_t fallback) const {
    const Value* v = find(key);
    return (v && v->is_number()) ? static_cast<int64_t>(v->number) : fallback;
  }
};

class Parser {
 public:
  explicit Parser(const std::string& text) : s_(text) {}

  Value parse() {
    Value v = value();
    skip_ws();
    if (pos_ != s_.size()) fail("trailing characters");
    return v;
  }

 private:
  [[noreturn]] void fail(const char* what) const {
    throw std::runtime_error("JSON parse error at byte " + std::to_string(pos_) + ": " + what);
  }
  void skip_ws() {
    while (pos_ < s_.size() && (s_[pos_] == ' ' || s_[pos_] == '\t' || s_[pos_] == '\n' || s_[pos_] == '\r')) ++pos_;
  }
  char peek() {
    skip_ws();
    if (pos_ >= s_.size()) fail("unexpected end");
    return s_[pos_];
  }
  bool consume(const char* lit) {
    size_t n = 0;
    while (lit[n]) ++n;
    if (s_.compare(pos_, n, lit) == 0) {
      pos_ += n;
      return true;
    }
    return false;
  }

  Value value() {
    if (++depth_ > 256) fail("nesting too deep");
    Value v;
    switch (peek()) {
      case '{': v = object(); break;
      case '[': v = array(); break;
      case '"': v.kind = Value::Kind::String; v.string = string(); break;
      case 't': if (!consume("true")) fail("bad literal"); v.kind = Value::Kind::Bool; v.boolean = true; break;
      case 'f': if (!consume("false")) fail("bad literal"); v.kind = Value::Kind::Bool; break;
      case 'n': if (!consume("null")) fail("bad literal"); break;
      default: v.kind = Value::Kind::Number; v.number = number(); break;
    }
    --depth_;
    return v;
  }

  Value object() {
    Value v;
    v.kind = Value::Kind::Object;
    ++pos_;
    if (peek() == '}') { ++pos_; return v; }
    for (;;) {
      if (peek() != '"') fail("object key expected");
      std::string key = string();
      if (peek() != ':') fail("':' expected");
      ++pos_;
      v.object.emplace_back(std::move(key), value());
      char c = peek();
      ++pos_;
      if (c == '}') return v;
      if (c != ',') fail("',' or '}' expected");
    }
  }

  Value array() {
    Value v;
    v.kind = Value::Kind::Array;
    ++pos_;
    if (peek() == ']') { ++pos_; return v; }
    for (;;) {
      v.array.push_back(value());
      char c = peek();
      ++pos_;
      if (c == ']') return v;
      if (c != ',') fail("',' or ']' expected");
    }
  }

  static void append_utf8(std::string& out, uint32_t cp) {
    if (cp < 0x80) out += static_cast<char>(cp);
    else if (cp < 0x800) { out += static_cast<char>(0xC0 | (cp >> 6)); out += static_cast<char>(0x80 | (cp & 0x3F)); }
    else if (cp < 0x10000) {
      out += static_cast<char>(0xE0 | (cp >> 12)); out += static_cast<char>(0x80 | ((cp >> 6) & 0x3F));
      out += static_cast<char>(0x80 | (cp & 0x3F));
    } else {
      out += static_cast<char>(0xF0 | (cp >> 18)); out += static_cast<char>(0x80 | ((cp >> 12) & 0x3F));
      out += static_cast<char>(0x80 | ((cp >> 6) & 0x3F)); out += static_cast<char>(0x80 | (cp & 0x3F));
    }
  }

  uint32_t hex4() {
    if (pos_ + 4 > s_.size()) fail("bad \\u escape");
    uint32_t v = 0;
    for (int i = 0; i < 4; ++i) {
      char c = s_[pos_++];
      v <<= 4;
      if (c >= '0' && c <= '9') v |= static_cast<uint32_t>(c - '0');
      else if (c >= 'a' && c <= 'f') v |= static_cast<uint32_t>(c - 'a' + 10);
      else if (c >= 'A' && c <= 'F') v |= static_cast<uint32_t>(c - 'A' + 10);
      else fail("bad \\u escape");
    }
    return v;
  }

  std::string string() {
    std::string out;
    ++pos_;  // opening quote
    for (;;) {
      if (pos_ >= s_.size()) fail("unterminated string");
      char c = s_[pos_++];
      if (c == '"') return out;
      if (c != '\\') { out += c; continue; }
      if (pos_ >= s_.size()) fail("unterminated escape");
      char e = s_[pos_++];
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
          uint32_t cp = hex4();
          if (cp >= 0xD800 && cp < 0xDC00 && s_.compare(pos_, 2, "\\u") == 0) {
            pos_ += 2;
            uint32_t lo = hex4();
            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
          }
          append_utf8(out, cp);
          break;
        }
        default: fail("bad escape");
      }
    }
  }

  double number() {
    const char* begin = s_.c_str() + pos_;
    char* end = nullptr;
    double v = std::strtod(begin, &end);
    if (end == begin) fail("value expected");
    pos_ += static_cast<size_t>(end - begin);
    return v;
  }

  const std::string& s_;
  size_t pos_ = 0;
  int depth_ = 0;
};

inline Value parse(const std::string& text) { return Parser(text).parse(); }

}  // namespace ptjson
