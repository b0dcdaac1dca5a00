// vti_reader.cpp -- see vti_reader.h.  Host-only C++; zlib is the one codec used (vtkZLibDataCompressor).
#include "vti_reader.h"

#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <sstream>

namespace dmi {
namespace host {
namespace vti {
namespace {

bool fail(std::string *err, const std::string &msg) {
  if (err) *err = msg;
  return false;
}

// value of name="..." inside the text of one XML tag
bool attr(const std::string &tag, const std::string &name, std::string *out) {
  const std::string key = name + "=\"";
  std::string::size_type p = tag.find(key);
  while (p != std::string::npos && p > 0 && !std::isspace((unsigned char)tag[p - 1])) p = tag.find(key, p + 1);
  if (p == std::string::npos) return false;
  const std::string::size_type b = p + key.size(), e = tag.find('"', b);
  if (e == std::string::npos) return false;
  *out = tag.substr(b, e - b);
  return true;
}

size_t type_size(const std::string &t) {
  if (t == "Int8" || t == "UInt8") return 1;
  if (t == "Int16" || t == "UInt16") return 2;
  if (t == "Int32" || t == "UInt32" || t == "Float32") return 4;
  if (t == "Int64" || t == "UInt64" || t == "Float64") return 8;
  return 0;
}

int b64_value(unsigned char c) {
  if (c >= 'A' && c <= 'Z') return c - 'A';
  if (c >= 'a' && c <= 'z') return c - 'a' + 26;
  if (c >= '0' && c <= '9') return c - '0' + 52;
  if (c == '+') return 62;
  if (c == '/') return 63;
  return -1;
}

// Decodes one base64 unit starting at text[*pos]: stops after `want` bytes (want == SIZE_MAX: until padding or a
// non-alphabet character), consuming whole 4-character groups; whitespace between groups is skipped.
bool b64_decode(const std::string &text, size_t *pos, size_t end, size_t want, std::vector<unsigned char> *out) {
  size_t p = *pos;
  while (out->size() < want) {
    int v[4];
    int got = 0, pad = 0;
    while (got < 4 && p < end) {
      const unsigned char c = (unsigned char)text[p];
      if (std::isspace(c)) {
        ++p;
        continue;
      }
      if (c == '=') {
        v[got++] = 0;
        ++pad;
        ++p;
        continue;
      }
      const int x = b64_value(c);
      if (x < 0) break;
      if (pad) return false;  // data after padding inside one group
      v[got++] = x;
      ++p;
    }
    if (got == 0) break;       // end of the unit
    if (got != 4) return false;
    const unsigned triple = (unsigned)(v[0] << 18 | v[1] << 12 | v[2] << 6 | v[3]);
    const int n = 3 - pad;
    if (n >= 1) out->push_back((unsigned char)(triple >> 16));
    if (n >= 2) out->push_back((unsigned char)(triple >> 8));
    if (n >= 3) out->push_back((unsigned char)triple);
    if (pad) break;  // a padded group ends the unit
  }
  *pos = p;
  return want == SIZE_MAX || out->size() >= want;
}

void swap_elements(unsigned char *p, size_t n_elems, size_t size) {
  if (size < 2) return;
  for (size_t i = 0; i < n_elems; ++i) std::reverse(p + i * size, p + (i + 1) * size);
}

struct Format {
  size_t header_word = 4;  // header_type UInt32 (the default of version 0.1 files) or UInt64
  bool swap = false;       // file byte order differs from the host's
  bool zlib = false;
};

uint64_t header_word(const unsigned char *p, const Format &f, size_t index) {
  unsigned char w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  std::memcpy(w, p + index * f.header_word, f.header_word);
  if (f.swap) std::reverse(w, w + f.header_word);
  uint64_t v = 0;
  std::memcpy(&v, w, 8);  // host is little-endian (x86-64); a swapped big-endian word is now little-endian
  if (f.header_word == 4) v &= 0xffffffffull;
  return v;
}

// `expect` = the byte count the extent and the array's type call for.  The extent is itself read from the file, so
// `expect` is bounded by what the compressed bytes actually present could inflate to (deflate's ratio never exceeds
// 1032 : 1) BEFORE anything is allocated: a tiny crafted file cannot ask for terabytes.
constexpr uint64_t kMaxInflateRatio = 1032;

bool inflate_blocks(const unsigned char *src, size_t src_len, const std::vector<uint64_t> &csize, uint64_t block,
                    uint64_t last, size_t expect, std::vector<unsigned char> *out, std::string *err) {
  const size_t nb = csize.size();
  // block is the nominal block size (an array smaller than one block has nb = 1, last = its size)
  if (last > block || (nb > 1 && block > expect) || nb > (uint64_t)expect + 1)
    return fail(err, "compression header does not match the array's size");
  uint64_t total = 0;
  for (size_t b = 0; b < nb; ++b) total += (b + 1 == nb && last != 0) ? last : block;
  if (total != expect) return fail(err, "compressed array does not have the size its extent and type call for");
  uint64_t cbytes = 0;
  for (size_t b = 0; b < nb; ++b) {
    if (csize[b] > src_len) return fail(err, "compressed block runs past the end of the data");
    cbytes += csize[b];
  }
  if (cbytes > src_len) return fail(err, "compressed blocks run past the end of the data");
  if (total > kMaxInflateRatio * cbytes + 64 * (uint64_t)nb)
    return fail(err, "the extent asks for more bytes than the compressed data can hold");
  out->resize((size_t)total);
  size_t in_off = 0, out_off = 0;
  for (size_t b = 0; b < nb; ++b) {
    const size_t want = (size_t)((b + 1 == nb && last != 0) ? last : block);
    if (csize[b] > src_len - in_off) return fail(err, "compressed block runs past the end of the data");
    uLongf got = (uLongf)want;
    const int rc = uncompress(out->data() + out_off, &got, src + in_off, (uLong)csize[b]);
    if (rc != Z_OK || got != want) return fail(err, "zlib: a compressed block does not inflate to its stated size");
    in_off += (size_t)csize[b];
    out_off += want;
  }
  return true;
}

// raw (not base64) payload at data[0 .. len)
bool decode_raw(const unsigned char *data, size_t len, const Format &f, size_t expect, std::vector<unsigned char> *out,
                std::string *err) {
  const size_t hw = f.header_word;
  if (!f.zlib) {
    if (len < hw) return fail(err, "appended data: truncated header");
    const uint64_t n = header_word(data, f, 0);
    if (n != expect) return fail(err, "appended data: array does not have the size its extent and type call for");
    if (n > len - hw) return fail(err, "appended data: array runs past the end of the file");
    out->assign(data + hw, data + hw + (size_t)n);
    return true;
  }
  if (len < 3 * hw) return fail(err, "appended data: truncated compression header");
  const uint64_t nb = header_word(data, f, 0), block = header_word(data, f, 1), last = header_word(data, f, 2);
  if (nb > (len - 3 * hw) / hw) return fail(err, "appended data: truncated compression header");
  std::vector<uint64_t> csize((size_t)nb);
  for (size_t b = 0; b < (size_t)nb; ++b) csize[b] = header_word(data, f, 3 + b);
  const size_t hbytes = (3 + (size_t)nb) * hw;
  return inflate_blocks(data + hbytes, len - hbytes, csize, block, last, expect, out, err);
}

// base64 payload starting at text[pos] (inline "binary" arrays and base64 appended data)
bool decode_b64(const std::string &text, size_t pos, size_t end, const Format &f, size_t expect,
                std::vector<unsigned char> *out, std::string *err) {
  const size_t hw = f.header_word;
  if (!f.zlib) {
    // one unit: [n_bytes] DATA
    std::vector<unsigned char> head;
    size_t p = pos;
    if (!b64_decode(text, &p, end, hw, &head)) return fail(err, "base64: truncated header");
    const uint64_t n = header_word(head.data(), f, 0);
    if (n != expect) return fail(err, "array does not have the size its extent and type call for");
    // four characters carry three bytes: the text that is left bounds the array before anything is reserved
    if (n > (uint64_t)(end - pos) / 4 * 3 + 3) return fail(err, "base64: array data is shorter than its header says");
    std::vector<unsigned char> all;
    all.reserve((size_t)n + hw + 3);
    p = pos;
    if (!b64_decode(text, &p, end, hw + (size_t)n, &all)) return fail(err, "base64: array data is shorter than its header says");
    out->assign(all.begin() + hw, all.begin() + hw + (size_t)n);
    return true;
  }
  // unit 1: the compression header; unit 2: the compressed blocks
  std::vector<unsigned char> head;
  size_t p = pos;
  if (!b64_decode(text, &p, end, 3 * hw, &head)) return fail(err, "base64: truncated compression header");
  const uint64_t nb = header_word(head.data(), f, 0), block = header_word(head.data(), f, 1), last = header_word(head.data(), f, 2);
  // every block but the last holds `block` bytes: more blocks than bytes cannot be
  if (nb > (uint64_t)expect + 1 || nb > (end - pos)) return fail(err, "compression header claims an absurd block count");
  const size_t hbytes = (3 + (size_t)nb) * hw;
  head.clear();
  p = pos;
  if (!b64_decode(text, &p, end, hbytes, &head)) return fail(err, "base64: truncated compression header");
  std::vector<uint64_t> csize((size_t)nb);
  uint64_t ctotal = 0;
  for (size_t b = 0; b < (size_t)nb; ++b) {
    csize[b] = header_word(head.data(), f, 3 + b);
    if (csize[b] > end - pos) return fail(err, "base64: a compressed block is larger than the file");
    ctotal += csize[b];
  }
  if (ctotal > end - pos) return fail(err, "base64: the compressed blocks are larger than the file");
  // the header unit is padded to whole groups: the data unit starts at the next group boundary
  size_t q = pos, groups = (hbytes + 2) / 3, seen = 0;
  while (seen < groups * 4 && q < end) {
    if (!std::isspace((unsigned char)text[q])) ++seen;
    ++q;
  }
  std::vector<unsigned char> comp;
  comp.reserve((size_t)ctotal + 3);
  if (!b64_decode(text, &q, end, (size_t)ctotal, &comp)) return fail(err, "base64: compressed data is shorter than its header says");
  return inflate_blocks(comp.data(), comp.size(), csize, block, last, expect, out, err);
}

}  // namespace

namespace {
bool read_image_data(const std::string &path, const std::vector<std::string> &wanted, Image *out, std::string *err);
}

bool ReadImageData(const std::string &path, const std::vector<std::string> &wanted, Image *out, std::string *err) {
  // nothing thrown by the containers (bad_alloc, length_error) may leave this function: its callers sit right below
  // extern "C" entry points
  try {
    return read_image_data(path, wanted, out, err);
  } catch (const std::exception &e) {
    return fail(err, path + ": " + e.what());
  } catch (...) {
    return fail(err, path + ": unknown failure while reading");
  }
}

namespace {
bool read_image_data(const std::string &path, const std::vector<std::string> &wanted, Image *out, std::string *err) {
  std::ifstream f(path.c_str(), std::ios::binary);
  if (!f.is_open()) return fail(err, "cannot open " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string text = ss.str();

  const std::string::size_type vf = text.find("<VTKFile");
  if (vf == std::string::npos) return fail(err, path + ": not a VTK XML file");
  const std::string vtag = text.substr(vf, text.find('>', vf) - vf);
  std::string s;
  if (!attr(vtag, "type", &s) || s != "ImageData") return fail(err, path + ": VTKFile type is not ImageData");
  Format fmt;
  if (attr(vtag, "header_type", &s)) {
    if (s == "UInt64") fmt.header_word = 8;
    else if (s != "UInt32") return fail(err, path + ": unsupported header_type " + s);
  }
  if (attr(vtag, "byte_order", &s)) fmt.swap = s == "BigEndian";
  if (attr(vtag, "compressor", &s) && !s.empty()) {
    if (s != "vtkZLibDataCompressor") return fail(err, path + ": unsupported compressor " + s + " (only vtkZLibDataCompressor)");
    fmt.zlib = true;
  }

  const std::string::size_type img = text.find("<ImageData", vf);
  if (img == std::string::npos) return fail(err, path + ": no <ImageData> element");
  const std::string itag = text.substr(img, text.find('>', img) - img);
  if (!attr(itag, "WholeExtent", &s)) return fail(err, path + ": <ImageData> without WholeExtent");
  {
    std::istringstream es(s);
    for (int i = 0; i < 6; ++i)
      if (!(es >> out->extent[i])) return fail(err, path + ": malformed WholeExtent");
  }
  if (attr(itag, "Origin", &s)) {
    std::istringstream os(s);
    for (int i = 0; i < 3; ++i) os >> out->origin[i];
  }
  if (attr(itag, "Spacing", &s)) {
    std::istringstream os(s);
    for (int i = 0; i < 3; ++i) os >> out->spacing[i];
  }
  for (int a = 0; a < 3; ++a)
    if (out->dims(a) < 1) return fail(err, path + ": empty extent");
  for (int a = 0; a < 3; ++a)
    if (out->dims(a) > (1 << 20)) return fail(err, path + ": extent too large");
  if ((double)out->dims(0) * out->dims(1) * out->dims(2) > 4e9) return fail(err, path + ": extent too large");
  const size_t n_points = (size_t)out->dims(0) * out->dims(1) * out->dims(2);

  // appended data section (raw payloads can contain anything: never search inside it)
  const std::string::size_type app = text.find("<AppendedData", img);
  size_t app_data = std::string::npos;
  bool app_raw = false;
  if (app != std::string::npos) {
    const std::string::size_type app_end = text.find('>', app);
    if (app_end == std::string::npos) return fail(err, path + ": malformed <AppendedData>");
    const std::string atag = text.substr(app, app_end - app);
    if (attr(atag, "encoding", &s)) app_raw = s == "raw";
    const std::string::size_type us = text.find('_', app_end);
    if (us == std::string::npos) return fail(err, path + ": <AppendedData> without the '_' marker");
    app_data = us + 1;
  }
  const size_t xml_end = app != std::string::npos ? app : text.size();

  std::string::size_type pd0 = text.find("<PointData", img), pd1 = std::string::npos;
  if (pd0 != std::string::npos && pd0 < xml_end) {
    const std::string::size_type tag_end = text.find('>', pd0);
    if (tag_end != std::string::npos && text[tag_end - 1] == '/') pd1 = tag_end;  // <PointData/>: no arrays
    else pd1 = text.find("</PointData>", pd0);
    if (pd1 == std::string::npos || pd1 > xml_end) return fail(err, path + ": unterminated <PointData>");
  } else {
    return fail(err, path + ": no <PointData> element");
  }

  out->point_data.clear();
  std::string::size_type p = pd0;
  while ((p = text.find("<DataArray", p)) != std::string::npos && p < pd1) {
    const std::string::size_type tag_end = text.find('>', p);
    if (tag_end == std::string::npos || tag_end > pd1) return fail(err, path + ": malformed <DataArray>");
    const bool self_closed = text[tag_end - 1] == '/';
    const std::string tag = text.substr(p, tag_end - p);
    p = tag_end + 1;
    Array a;
    attr(tag, "Name", &a.name);
    if (!wanted.empty() && std::find(wanted.begin(), wanted.end(), a.name) == wanted.end()) continue;
    std::string format = "ascii";
    attr(tag, "format", &format);
    if (!attr(tag, "type", &a.type) || (a.elem_size = type_size(a.type)) == 0)
      return fail(err, path + ": array '" + a.name + "' has an unknown type");
    if (attr(tag, "NumberOfComponents", &s)) a.components = std::atoi(s.c_str());
    if (a.components < 1 || a.components > 1024) return fail(err, path + ": array '" + a.name + "' has an impossible component count");
    const size_t n_values = n_points * (size_t)a.components;
    const size_t n_bytes = n_values * a.elem_size;

    if (format == "appended") {
      if (app_data == std::string::npos) return fail(err, path + ": appended array without <AppendedData>");
      if (!attr(tag, "offset", &s)) return fail(err, path + ": appended array '" + a.name + "' without offset");
      const unsigned long long off = std::strtoull(s.c_str(), nullptr, 10);
      if (off > text.size() - app_data) return fail(err, path + ": offset of '" + a.name + "' is past the end of the file");
      std::string why;
      const bool ok = app_raw ? decode_raw(reinterpret_cast<const unsigned char *>(text.data()) + app_data + off,
                                           text.size() - app_data - (size_t)off, fmt, n_bytes, &a.bytes, &why)
                              : decode_b64(text, app_data + (size_t)off, text.size(), fmt, n_bytes, &a.bytes, &why);
      if (!ok) return fail(err, path + ": array '" + a.name + "': " + why);
    } else {
      if (self_closed) return fail(err, path + ": inline array '" + a.name + "' has no content");
      const std::string::size_type close = text.find("</DataArray>", p);
      if (close == std::string::npos || close > pd1) return fail(err, path + ": unterminated <DataArray>");
      if (format == "binary") {
        std::string why;
        if (!decode_b64(text, p, close, fmt, n_bytes, &a.bytes, &why)) return fail(err, path + ": array '" + a.name + "': " + why);
      } else if (format == "ascii") {
        if (n_values > (close - p)) return fail(err, path + ": array '" + a.name + "' has fewer values than points");
        a.bytes.resize(n_bytes);
        const char *c = text.data() + p;
        const char *const cend = text.data() + close;
        const bool is_float = a.type == "Float32" || a.type == "Float64";
        const bool is_signed = a.type[0] == 'I';
        for (size_t i = 0; i < n_values; ++i) {
          while (c < cend && std::isspace((unsigned char)*c)) ++c;
          if (c >= cend) return fail(err, path + ": array '" + a.name + "' has fewer values than points");
          char *next = nullptr;
          unsigned char *dst = a.bytes.data() + i * a.elem_size;
          if (is_float) {
            const double v = std::strtod(c, &next);
            if (a.elem_size == 8) std::memcpy(dst, &v, 8);
            else { const float w = (float)v; std::memcpy(dst, &w, 4); }
          } else if (is_signed) {
            const long long v = std::strtoll(c, &next, 10);
            std::memcpy(dst, &v, a.elem_size);  // little-endian host: the low bytes
          } else {
            const unsigned long long v = std::strtoull(c, &next, 10);
            std::memcpy(dst, &v, a.elem_size);
          }
          if (next == c) return fail(err, path + ": array '" + a.name + "' holds something that is not a number");
          c = next;
        }
        p = close;
        out->point_data.push_back(std::move(a));
        continue;  // text values have no byte order
      } else {
        return fail(err, path + ": array '" + a.name + "' has unknown format " + format);
      }
      p = close;
    }
    if (a.bytes.size() != n_bytes)
      return fail(err, path + ": array '" + a.name + "' holds " + std::to_string(a.bytes.size()) + " bytes, the extent needs " +
                           std::to_string(n_bytes));
    if (fmt.swap) swap_elements(a.bytes.data(), n_values, a.elem_size);
    out->point_data.push_back(std::move(a));
  }
  return true;
}
}  // namespace

}  // namespace vti
}  // namespace host
}  // namespace dmi
