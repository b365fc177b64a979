// glTF 2.0 import on the host side of the C ABI: pt_gltf_load produces the flat arrays that the reference's
// Scene::load hands to its renderers (src/scene.cpp:56-155: loadGltfScene through tinygltf, then the un-vendored
// nvh::GltfScene::importMaterials / importDrawableNodes(Normal | Texcoord_0 | Tangent | Color_0), then
// createMaterialBuffer :339-382, createLightBuffer :304-333, createTextureImages :488-580, createVertexBuffer :190-274,
// setCameraFromScene :281-298) -- ready for pt_set_scene.  No third-party code: a small JSON reader, base64, the GLB
// container, PNG through zlib's inflate and a baseline JPEG decoder live in this file.
//
// What nvh::GltfScene does where a file leaves attributes out is not in the reference tree (parity unpinned); this
// importer follows the glTF 2.0 specification: missing normals -> area-weighted vertex normals, missing tangents ->
// per-vertex tangents from the uv parameterisation (handedness in w), missing uv -> 0, missing colour -> 1.
// vk_raytrace_amd/gltf.py is the same importer in Python; tests/test_gltf_cpp.py holds the two to identical output.
#include <zlib.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>
#include "../../include/pt_api.h"

namespace {

[[noreturn]] void fail(const char* fmt, ...)
{
  char    buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  throw std::runtime_error(buf);
}

// ---------------------------------------------------------------------------------------------- JSON
struct JVal {
  enum Type { Null, Bool, Num, Str, Arr, Obj } t = Null;
  double                                    num = 0;
  bool                                      b   = false;
  std::string                               str;
  std::vector<JVal>                         arr;
  std::vector<std::pair<std::string, JVal>> obj;

  const JVal* get(const char* k) const
  {
    if(t != Obj)
      return nullptr;
    for(auto& kv : obj)
      if(kv.first == k)
        return &kv.second;
    return nullptr;
  }
  bool        has(const char* k) const { return get(k) != nullptr; }
  double      number(const char* k, double def) const { const JVal* v = get(k); return (v && v->t == Num) ? v->num : def; }
  int         integer(const char* k, int def) const { const JVal* v = get(k); return (v && v->t == Num) ? int(v->num) : def; }
  bool        boolean(const char* k, bool def) const { const JVal* v = get(k); return (v && v->t == Bool) ? v->b : def; }
  std::string string(const char* k, const char* def) const { const JVal* v = get(k); return (v && v->t == Str) ? v->str : std::string(def); }
  const std::vector<JVal>& array(const char* k) const
  {
    static const std::vector<JVal> empty;
    const JVal*                    v = get(k);
    return (v && v->t == Arr) ? v->arr : empty;
  }
};

struct JParser {
  const char* p;
  const char* end;
  int         nest = 0;  // current nesting depth: the parser recurses per container, an adversarial file must not exhaust the stack
  void        ws() { while(p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p; }
  struct Nest {
    int& n;
    explicit Nest(int& d) : n(d)
    {
      if(++n > 256)
        fail("JSON: nesting deeper than 256 levels");
    }
    ~Nest() { --n; }
  };
  JVal parse()
  {
    Nest guard(nest);
    ws();
    if(p >= end)
      fail("JSON: unexpected end");
    JVal v;
    if(*p == '{')
    {
      v.t = JVal::Obj;
      ++p;
      ws();
      if(p < end && *p == '}') { ++p; return v; }
      for(;;)
      {
        ws();
        std::string k = str();
        ws();
        if(p >= end || *p != ':')
          fail("JSON: ':' expected");
        ++p;
        v.obj.emplace_back(std::move(k), parse());
        ws();
        if(p < end && *p == ',') { ++p; continue; }
        if(p < end && *p == '}') { ++p; return v; }
        fail("JSON: ',' or '}' expected");
      }
    }
    if(*p == '[')
    {
      v.t = JVal::Arr;
      ++p;
      ws();
      if(p < end && *p == ']') { ++p; return v; }
      for(;;)
      {
        v.arr.push_back(parse());
        ws();
        if(p < end && *p == ',') { ++p; continue; }
        if(p < end && *p == ']') { ++p; return v; }
        fail("JSON: ',' or ']' expected");
      }
    }
    if(*p == '"') { v.t = JVal::Str; v.str = str(); return v; }
    if(end - p >= 4 && !strncmp(p, "true", 4)) { p += 4; v.t = JVal::Bool; v.b = true; return v; }
    if(end - p >= 5 && !strncmp(p, "false", 5)) { p += 5; v.t = JVal::Bool; v.b = false; return v; }
    if(end - p >= 4 && !strncmp(p, "null", 4)) { p += 4; return v; }
    char* e = nullptr;
    v.num   = strtod(p, &e);
    if(e == p)
      fail("JSON: value expected");
    v.t = JVal::Num;
    p   = e;
    return v;
  }
  std::string str()
  {
    if(p >= end || *p != '"')
      fail("JSON: string expected");
    ++p;
    std::string s;
    while(p < end && *p != '"')
    {
      if(*p == '\\' && p + 1 < end)
      {
        ++p;
        switch(*p)
        {
          case 'n': s += '\n'; break;
          case 't': s += '\t'; break;
          case 'r': s += '\r'; break;
          case 'b': s += '\b'; break;
          case 'f': s += '\f'; break;
          case 'u': {
            if(end - p < 5)
              fail("JSON: bad \\u escape");
            unsigned cp = unsigned(strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16));
            p += 4;
            if(cp < 0x80) s += char(cp);
            else if(cp < 0x800) { s += char(0xC0 | (cp >> 6)); s += char(0x80 | (cp & 0x3F)); }
            else { s += char(0xE0 | (cp >> 12)); s += char(0x80 | ((cp >> 6) & 0x3F)); s += char(0x80 | (cp & 0x3F)); }
            break;
          }
          default: s += *p;
        }
        ++p;
      }
      else
        s += *p++;
    }
    if(p >= end)
      fail("JSON: unterminated string");
    ++p;
    return s;
  }
};

// ---------------------------------------------------------------------------------------------- bytes
typedef std::vector<uint8_t> Bytes;

Bytes read_file(const std::string& path)
{
  FILE* f = fopen(path.c_str(), "rb");
  if(!f)
    fail("cannot open %s", path.c_str());
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  Bytes b(size_t(n > 0 ? n : 0));
  if(n > 0 && fread(b.data(), 1, size_t(n), f) != size_t(n))
  {
    fclose(f);
    fail("short read on %s", path.c_str());
  }
  fclose(f);
  return b;
}

Bytes base64(const std::string& s, size_t from)
{
  Bytes    out;
  uint32_t acc = 0;
  int      bits = 0;
  for(size_t i = from; i < s.size(); ++i)
  {
    char c = s[i];
    int  v;
    if(c >= 'A' && c <= 'Z') v = c - 'A';
    else if(c >= 'a' && c <= 'z') v = c - 'a' + 26;
    else if(c >= '0' && c <= '9') v = c - '0' + 52;
    else if(c == '+' || c == '-') v = 62;
    else if(c == '/' || c == '_') v = 63;
    else continue;  // padding / whitespace
    acc = (acc << 6) | uint32_t(v);
    bits += 6;
    if(bits >= 8)
    {
      bits -= 8;
      out.push_back(uint8_t((acc >> bits) & 0xFF));
    }
  }
  return out;
}

Bytes read_uri(const std::string& uri, const std::string& base)
{
  if(uri.compare(0, 5, "data:") == 0)
  {
    size_t comma = uri.find(',');
    if(comma == std::string::npos || uri.substr(0, comma).find(";base64") == std::string::npos)
      fail("data: uris must be base64");
    return base64(uri, comma + 1);
  }
  std::string path;
  for(size_t i = 0; i < uri.size(); ++i)  // percent-decoding
    if(uri[i] == '%' && i + 2 < uri.size())
    {
      path += char(strtoul(uri.substr(i + 1, 2).c_str(), nullptr, 16));
      i += 2;
    }
    else
      path += uri[i];
  return read_file(base.empty() ? path : base + "/" + path);
}

// ---------------------------------------------------------------------------------------------- images
struct Image {
  int   w = 0, h = 0;
  Bytes rgba;
};

uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

Image decode_png(const Bytes& d)
{
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if(d.size() < 8 || memcmp(d.data(), sig, 8))
    fail("PNG: bad signature");
  int     w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
  Bytes   idat, plte, trns;
  size_t  off = 8;
  while(off + 12 <= d.size())
  {
    uint32_t    len  = be32(&d[off]);
    const char* type = (const char*)&d[off + 4];
    if(off + 12 + len > d.size())
      fail("PNG: truncated chunk");
    const uint8_t* body = &d[off + 8];
    if(!memcmp(type, "IHDR", 4))
    {
      if(len < 13)
        fail("PNG: short IHDR");
      w = int(be32(body)); h = int(be32(body + 4)); depth = body[8]; ctype = body[9]; interlace = body[12];
    }
    else if(!memcmp(type, "PLTE", 4)) plte.assign(body, body + len);
    else if(!memcmp(type, "tRNS", 4)) trns.assign(body, body + len);
    else if(!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
    else if(!memcmp(type, "IEND", 4)) break;
    off += 12 + len;
  }
  if(w <= 0 || h <= 0)
    fail("PNG: no IHDR");
  if(interlace > 1)
    fail("PNG: unknown interlace method %d", interlace);
  const int ch  = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  // PNG 1.2 table 11.1: allowed bit depths per colour type
  const bool depthOk = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) || (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8))
                       || ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
  if(!ch || !depthOk)
    fail("PNG: unsupported colour type %d / depth %d", ctype, depth);
  if(size_t(w) > 32768 || size_t(h) > 32768)
    fail("PNG: %d x %d is larger than 32768 x 32768", w, h);
  const size_t bpp = std::max<size_t>(1, size_t(ch) * depth / 8);  // bytes per complete pixel for filtering
  // (x0, y0, dx, dy) of the passes: one for a plain image, Adam7's seven for an interlaced one
  static const int adam7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
  static const int plain[1][4] = {{0, 0, 1, 1}};
  const int(*passes)[4]        = interlace ? adam7 : plain;
  const int npass              = interlace ? 7 : 1;
  auto      pass_dim = [&](int k, int& pw, int& ph) {
    pw = (w - passes[k][0] + passes[k][2] - 1) / passes[k][2];
    ph = (h - passes[k][1] + passes[k][3] - 1) / passes[k][3];
    if(pw < 0) pw = 0;
    if(ph < 0) ph = 0;
  };
  size_t total = 0;
  for(int k = 0; k < npass; ++k)
  {
    int pw, ph;
    pass_dim(k, pw, ph);
    if(pw && ph)
      total += ((size_t(pw) * ch * depth + 7) / 8 + 1) * size_t(ph);
  }
  Bytes  raw(total);
  uLongf rawLen = uLongf(raw.size());
  if(uncompress(raw.data(), &rawLen, idat.data(), uLong(idat.size())) != Z_OK || rawLen != raw.size())
    fail("PNG: inflate failed");
  Image im;
  im.w = w; im.h = h;
  im.rgba.resize(size_t(w) * h * 4);
  size_t rawOff = 0;
  for(int k = 0; k < npass; ++k)
  {
    int pw, ph;
    pass_dim(k, pw, ph);
    if(!pw || !ph)
      continue;
    const size_t stride = (size_t(pw) * ch * depth + 7) / 8;
    Bytes        prev(stride, 0), cur(stride);
    for(int y = 0; y < ph; ++y)
    {
      const uint8_t* row = &raw[rawOff + (stride + 1) * size_t(y)];
      const int      ft  = row[0];
      for(size_t i = 0; i < stride; ++i)
      {
        const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
        int       x = row[1 + i];
        switch(ft)
        {
          case 0: break;
          case 1: x += a; break;
          case 2: x += b; break;
          case 3: x += (a + b) >> 1; break;
          case 4: {
            int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
            x += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            break;
          }
          default: fail("PNG: bad filter %d", ft);
        }
        cur[i] = uint8_t(x);
      }
      for(int x = 0; x < pw; ++x)
      {
        auto sample = [&](int q) -> int {  // q-th channel of pixel x as 8 bit
          if(depth == 8) return cur[size_t(x) * ch + q];
          if(depth == 16) return cur[(size_t(x) * ch + q) * 2];  // high byte
          const int per = 8 / depth, byte = x / per, sh = (per - 1 - x % per) * depth;
          const int v = (cur[byte] >> sh) & ((1 << depth) - 1);
          return ctype == 3 ? v : v * 255 / ((1 << depth) - 1);
        };
        int r, g, b, a = 255;
        // tRNS of grey / RGB images: one 16-bit big-endian sample value per channel, compared with the RAW sample of the pixel
        auto raw = [&](int q) -> int {
          if(depth == 8) return cur[size_t(x) * ch + q];
          if(depth == 16) return (cur[(size_t(x) * ch + q) * 2] << 8) | cur[(size_t(x) * ch + q) * 2 + 1];
          const int per = 8 / depth, byte = x / per, sh = (per - 1 - x % per) * depth;
          return (cur[byte] >> sh) & ((1 << depth) - 1);
        };
        auto key = [&](int q) -> int { return (trns[size_t(q) * 2] << 8) | trns[size_t(q) * 2 + 1]; };
        if(ctype == 0) { r = g = b = sample(0); if(trns.size() >= 2 && raw(0) == key(0)) a = 0; }
        else if(ctype == 2) { r = sample(0); g = sample(1); b = sample(2); if(trns.size() >= 6 && raw(0) == key(0) && raw(1) == key(1) && raw(2) == key(2)) a = 0; }
        else if(ctype == 3)
        {
          const int i = sample(0);
          if(size_t(i) * 3 + 2 >= plte.size())
            fail("PNG: palette index out of range");
          r = plte[i * 3]; g = plte[i * 3 + 1]; b = plte[i * 3 + 2];
          a = size_t(i) < trns.size() ? trns[i] : 255;
        }
        else if(ctype == 4) { r = g = b = sample(0); a = sample(1); }
        else { r = sample(0); g = sample(1); b = sample(2); a = sample(3); }
        uint8_t* o = &im.rgba[(size_t(passes[k][1] + y * passes[k][3]) * w + size_t(passes[k][0] + x * passes[k][2])) * 4];
        o[0] = uint8_t(r); o[1] = uint8_t(g); o[2] = uint8_t(b); o[3] = uint8_t(a);
      }
      prev.swap(cur);
    }
    rawOff += (stride + 1) * size_t(ph);
  }
  return im;
}

// Huffman-coded 8-bit JPEG: baseline / extended sequential (SOF0, SOF1) and progressive (SOF2), interleaved or not, restart intervals.
// Every scan decodes into per-component coefficient planes; dequantisation, IDCT and colour conversion run once at the end.  Chroma is
// upsampled by replication and the IDCT is the separable float one -- decoders legitimately differ by an LSB or so (SURVEY.md 8(c):
// JPEG decode is unpinned).
struct Jpeg {
  const uint8_t* p;
  const uint8_t* end;
  struct Huff {
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    int     mincode[17], maxcode[18], valptr[17];
    bool    set = false;
    void    build()
    {
      int code = 0, k = 0;
      for(int l = 1; l <= 16; ++l)
      {
        valptr[l]  = k;
        mincode[l] = code;
        code += bits[l];
        k += bits[l];
        maxcode[l] = bits[l] ? code - 1 : -1;
        code <<= 1;
      }
      maxcode[17] = 0x7fffffff;
      set         = true;
    }
  } dc[4], ac[4];
  uint16_t qt[4][64] = {{0}};
  struct Comp {
    int                  id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0;
    int                  bw = 0, bh = 0;  // blocks per row / column, padded to whole MCUs
    int                  cw = 0, ch = 0;  // blocks that cover the component (non-interleaved scans visit only these)
    std::vector<int16_t> coef;            // bw * bh * 64, natural (de-zigzagged) order
    std::vector<uint8_t> plane;
  } comp[3];
  int      ncomp = 0, W = 0, H = 0, restart = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
  bool     progressive = false, haveFrame = false;
  uint32_t bitbuf = 0;
  int      bitcnt = 0;
  bool     hitMarker = false;
  int      eobrun = 0;

  int getbit()
  {
    if(bitcnt == 0)
    {
      int c = 0;
      if(p < end && !hitMarker)
      {
        c = *p++;
        if(c == 0xFF)
        {
          int c2 = p < end ? *p : 0;
          if(c2 == 0) ++p;
          else { hitMarker = true; --p; c = 0; }  // a marker: feed zeros until the caller consumes it
        }
      }
      bitbuf = uint32_t(c);
      bitcnt = 8;
    }
    return (bitbuf >> --bitcnt) & 1;
  }
  int getbits(int n)
  {
    if(n < 0 || n > 16)
      fail("JPEG: bad bit count");
    unsigned v = 0;
    while(n--)
      v = (v << 1) | unsigned(getbit());
    return int(v);
  }
  int decode(const Huff& h)
  {
    if(!h.set)
      fail("JPEG: missing Huffman table");
    int code = 0;
    for(int l = 1; l <= 16; ++l)
    {
      code = (code << 1) | getbit();
      if(h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l])
        return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    fail("JPEG: bad Huffman code");
  }
  static int extend(int v, int t) { return (t > 0 && v < (1 << (t - 1))) ? v - (1 << t) + 1 : v; }

  static const uint8_t* zigzag()
  {
    static const uint8_t zz[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    return zz;
  }

  static void idct8x8(const float* in, uint8_t* out, int stride)
  {
    static float c[8][8];
    static bool  init = false;
    if(!init)
    {
      for(int x = 0; x < 8; ++x)
        for(int u = 0; u < 8; ++u)
          c[x][u] = float((u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * M_PI / 16.0));
      init = true;
    }
    float tmp[64];
    for(int y = 0; y < 8; ++y)
      for(int x = 0; x < 8; ++x)
      {
        float s = 0;
        for(int u = 0; u < 8; ++u)
          s += c[x][u] * in[y * 8 + u];
        tmp[y * 8 + x] = s;
      }
    for(int x = 0; x < 8; ++x)
      for(int y = 0; y < 8; ++y)
      {
        float s = 0;
        for(int v = 0; v < 8; ++v)
          s += c[y][v] * tmp[v * 8 + x];
        int q = int(std::lrintf(s + 128.0f));
        out[y * stride + x] = uint8_t(q < 0 ? 0 : (q > 255 ? 255 : q));
      }
  }

  // ---- one 8x8 block of a scan
  void block_sequential(Comp& c, int16_t* co)
  {
    const uint8_t* zz = zigzag();
    const int      t  = decode(dc[c.td & 3]);
    if(t > 15)
      fail("JPEG: bad DC size");
    c.pred += t ? extend(getbits(t), t) : 0;
    co[0] = int16_t(c.pred);
    for(int k = 1; k < 64;)
    {
      const int rs = decode(ac[c.ta & 3]), r = rs >> 4, sz = rs & 15;
      if(sz == 0)
      {
        if(r != 15) break;
        k += 16;
        continue;
      }
      k += r;
      if(k > 63)
        fail("JPEG: bad coefficient index");
      co[zz[k]] = int16_t(extend(getbits(sz), sz));
      ++k;
    }
  }
  void block_dc_progressive(Comp& c, int16_t* co, int ah, int al)
  {
    if(ah == 0)
    {
      const int t = decode(dc[c.td & 3]);
      if(t > 15)
        fail("JPEG: bad DC size");
      c.pred += t ? extend(getbits(t), t) : 0;
      co[0] = int16_t(c.pred * (1 << al));
    }
    else if(getbit())
      co[0] = int16_t(co[0] | (1 << al));
  }
  void block_ac_progressive(Comp& c, int16_t* co, int ss, int se, int ah, int al)
  {
    const uint8_t* zz = zigzag();
    if(ah == 0)
    {
      if(eobrun > 0) { --eobrun; return; }
      for(int k = ss; k <= se;)
      {
        const int rs = decode(ac[c.ta & 3]), r = rs >> 4, sz = rs & 15;
        if(sz == 0)
        {
          if(r < 15)
          {
            eobrun = (1 << r) - 1;
            if(r) eobrun += getbits(r);
            break;
          }
          k += 16;
        }
        else
        {
          k += r;
          if(k > 63)
            fail("JPEG: bad coefficient index");
          co[zz[k]] = int16_t(extend(getbits(sz), sz) * (1 << al));
          ++k;
        }
      }
      return;
    }
    const int bit = 1 << al;
    auto      refine = [&](int16_t& v) {
      if(getbit() && (v & bit) == 0)
        v = int16_t(v > 0 ? v + bit : v - bit);
    };
    if(eobrun > 0)
    {
      --eobrun;
      for(int k = ss; k <= se; ++k)
        if(co[zz[k]] != 0)
          refine(co[zz[k]]);
      return;
    }
    int k = ss;
    while(k <= se)
    {
      const int rs = decode(ac[c.ta & 3]);
      int       r = rs >> 4, sz = rs & 15, val = 0;
      if(sz == 0)
      {
        if(r < 15)
        {
          eobrun = (1 << r) - 1;
          if(r) eobrun += getbits(r);
          r = 64;  // the rest of the band is refinement only
        }
      }
      else
      {
        if(sz != 1)
          fail("JPEG: bad refinement code");
        val = getbit() ? bit : -bit;
      }
      while(k <= se)
      {
        int16_t& v = co[zz[k++]];
        if(v != 0)
          refine(v);
        else
        {
          if(r == 0)
          {
            if(val) v = int16_t(val);
            break;
          }
          --r;
        }
      }
    }
  }

  void restart_marker()
  {
    bitcnt    = 0;
    hitMarker = false;
    while(p + 1 < end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) ++p;
    if(p + 1 < end) p += 2;
    for(int i = 0; i < ncomp; ++i) comp[i].pred = 0;
    eobrun = 0;
  }

  void scan(int ns, const int* ci, int ss, int se, int ah, int al)
  {
    bitcnt = 0; hitMarker = false; eobrun = 0;
    for(int i = 0; i < ncomp; ++i) comp[i].pred = 0;
    auto one = [&](Comp& c, int bx, int by) {
      int16_t* co = &c.coef[(size_t(by) * c.bw + bx) * 64];
      if(!progressive) block_sequential(c, co);
      else if(ss == 0) block_dc_progressive(c, co, ah, al);
      else block_ac_progressive(c, co, ss, se, ah, al);
    };
    int count = 0;
    if(ns == 1)
    {  // non-interleaved: the blocks that cover the component, in raster order
      Comp& c = comp[ci[0]];
      for(int by = 0; by < c.ch; ++by)
        for(int bx = 0; bx < c.cw; ++bx)
        {
          if(restart && count && count % restart == 0) restart_marker();
          ++count;
          one(c, bx, by);
        }
    }
    else
    {
      if(progressive && ss != 0)
        fail("JPEG: progressive AC scans must not be interleaved");
      for(int my = 0; my < mcuy; ++my)
        for(int mx = 0; mx < mcux; ++mx)
        {
          if(restart && count && count % restart == 0) restart_marker();
          ++count;
          for(int i = 0; i < ns; ++i)
          {
            Comp& c = comp[ci[i]];
            for(int by = 0; by < c.v; ++by)
              for(int bx = 0; bx < c.h; ++bx)
                one(c, mx * c.h + bx, my * c.v + by);
          }
        }
    }
    // position on the next marker
    bitcnt = 0;
    while(p + 1 < end && !(p[0] == 0xFF && p[1] != 0 && !(p[1] >= 0xD0 && p[1] <= 0xD7))) ++p;
    hitMarker = false;
  }

  Image run()
  {
    const uint8_t* zz = zigzag();
    if(end - p < 2 || p[0] != 0xFF || p[1] != 0xD8)
      fail("JPEG: no SOI");
    p += 2;
    bool scanned = false;
    for(;;)
    {
      if(end - p < 2)
        break;  // tolerate a missing EOI
      if(p[0] != 0xFF)
        fail("JPEG: marker expected");
      while(p < end && *p == 0xFF) ++p;
      if(p >= end)
        break;
      const int m = *p++;
      if(m == 0xD9)
        break;
      if(end - p < 2)
        fail("JPEG: truncated segment");
      const int      len = (p[0] << 8) | p[1];
      const uint8_t* s   = p + 2;
      const uint8_t* se  = p + len;
      if(se > end || len < 2)
        fail("JPEG: truncated segment");
      p = se;
      if(m == 0xDB)
        while(s < se)
        {
          const int pq = *s >> 4, tq = *s & 15;
          ++s;
          if(se - s < (pq ? 128 : 64))
            fail("JPEG: truncated quantisation table");
          for(int i = 0; i < 64; ++i)
          {
            qt[tq & 3][zz[i]] = pq ? uint16_t((s[0] << 8) | s[1]) : *s;
            s += pq ? 2 : 1;
          }
        }
      else if(m == 0xC0 || m == 0xC1 || m == 0xC2)
      {
        progressive = m == 0xC2;
        if(se - s < 6 || se - s < 6 + 3 * s[5])
          fail("JPEG: truncated frame header");
        if(haveFrame)
          fail("JPEG: more than one frame");
        if(s[0] != 8)
          fail("JPEG: only 8-bit samples are supported");
        H = (s[1] << 8) | s[2]; W = (s[3] << 8) | s[4]; ncomp = s[5];
        if(ncomp != 1 && ncomp != 3)
          fail("JPEG: %d components are not supported", ncomp);
        if(W <= 0 || H <= 0)
          fail("JPEG: empty frame");
        for(int i = 0; i < ncomp; ++i)
        {
          comp[i].id = s[6 + i * 3]; comp[i].h = s[7 + i * 3] >> 4; comp[i].v = s[7 + i * 3] & 15; comp[i].tq = s[8 + i * 3] & 3;
          if(comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4)
            fail("JPEG: bad sampling factors");
          hmax = std::max(hmax, comp[i].h); vmax = std::max(vmax, comp[i].v);
        }
        mcux = (W + 8 * hmax - 1) / (8 * hmax); mcuy = (H + 8 * vmax - 1) / (8 * vmax);
        for(int i = 0; i < ncomp; ++i)
        {
          Comp& c = comp[i];
          c.bw = mcux * c.h; c.bh = mcuy * c.v;
          c.cw = ((W * c.h + hmax - 1) / hmax + 7) / 8; c.ch = ((H * c.v + vmax - 1) / vmax + 7) / 8;
          c.coef.assign(size_t(c.bw) * c.bh * 64, 0);
        }
        haveFrame = true;
      }
      else if(m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)
        fail("JPEG: lossless / arithmetic-coded files are not supported (SOF%d)", m - 0xC0);
      else if(m == 0xC4)
        while(s < se)
        {
          const int tc = *s >> 4, th = *s & 3;
          ++s;
          Huff& h = tc ? ac[th] : dc[th];
          int   n = 0;
          if(se - s < 16)
            fail("JPEG: truncated Huffman table");
          for(int l = 1; l <= 16; ++l) { h.bits[l] = s[l - 1]; n += h.bits[l]; }
          s += 16;
          if(n > 256 || se - s < n)
            fail("JPEG: bad DHT");
          memcpy(h.vals, s, size_t(n));
          s += n;
          h.build();
        }
      else if(m == 0xDD)
      {
        if(se - s < 2)
          fail("JPEG: truncated DRI");
        restart = (s[0] << 8) | s[1];
      }
      else if(m == 0xDA)
      {
        if(!haveFrame)
          fail("JPEG: scan before the frame header");
        if(se - s < 1)
          fail("JPEG: truncated scan header");
        const int ns = s[0];
        if(ns < 1 || ns > ncomp || se - s < 1 + ns * 2)
          fail("JPEG: bad scan header");
        int ci[3] = {0, 0, 0};
        for(int i = 0; i < ns; ++i)
        {
          int k = 0;
          while(k < ncomp && comp[k].id != s[1 + i * 2]) ++k;
          if(k == ncomp)
            fail("JPEG: scan references an unknown component");
          comp[k].td = s[2 + i * 2] >> 4;
          comp[k].ta = s[2 + i * 2] & 15;
          ci[i]      = k;
        }
        if(se - s < 4 + ns * 2)
          fail("JPEG: truncated scan header");
        const int ss = s[1 + ns * 2], sen = s[2 + ns * 2], ah = s[3 + ns * 2] >> 4, al = s[3 + ns * 2] & 15;
        if(progressive && (ss > sen || sen > 63 || al > 13 || ah > 13))
          fail("JPEG: bad spectral selection");
        scan(ns, ci, progressive ? ss : 0, progressive ? sen : 63, progressive ? ah : 0, progressive ? al : 0);
        scanned = true;
      }
    }
    if(!haveFrame || !scanned)
      fail("JPEG: no image data");
    float blk[64];
    for(int i = 0; i < ncomp; ++i)
    {
      Comp& c = comp[i];
      c.plane.assign(size_t(c.bw) * 8 * size_t(c.bh) * 8, 0);
      const uint16_t* q = qt[c.tq];
      for(int by = 0; by < c.bh; ++by)
        for(int bx = 0; bx < c.bw; ++bx)
        {
          const int16_t* co = &c.coef[(size_t(by) * c.bw + bx) * 64];
          for(int k = 0; k < 64; ++k)
            blk[k] = float(int(co[k]) * int(q[k]));
          idct8x8(blk, &c.plane[size_t(by) * 8 * (size_t(c.bw) * 8) + size_t(bx) * 8], c.bw * 8);
        }
    }
    Image im;
    im.w = W; im.h = H;
    im.rgba.resize(size_t(W) * H * 4);
    for(int y = 0; y < H; ++y)
      for(int x = 0; x < W; ++x)
      {
        auto at = [&](int i) { return int(comp[i].plane[size_t(y * comp[i].v / vmax) * (size_t(comp[i].bw) * 8) + size_t(x * comp[i].h / hmax)]); };
        int  r, g, b;
        if(ncomp == 1)
          r = g = b = at(0);
        else
        {
          const float Y = float(at(0)), cb = float(at(1)) - 128.0f, cr = float(at(2)) - 128.0f;
          auto        cl = [](float v) { int q = int(std::lrintf(v)); return q < 0 ? 0 : (q > 255 ? 255 : q); };
          r = cl(Y + 1.402f * cr); g = cl(Y - 0.344136f * cb - 0.714136f * cr); b = cl(Y + 1.772f * cb);
        }
        uint8_t* o = &im.rgba[(size_t(y) * W + x) * 4];
        o[0] = uint8_t(r); o[1] = uint8_t(g); o[2] = uint8_t(b); o[3] = 255;
      }
    return im;
  }
};

Image decode_image(const Bytes& d, const char* what)
{
  if(d.size() >= 8 && d[0] == 0x89 && d[1] == 'P')
    return decode_png(d);
  if(d.size() >= 3 && d[0] == 0xFF && d[1] == 0xD8)
  {
    Jpeg j;
    j.p = d.data(); j.end = d.data() + d.size();
    return j.run();
  }
  fail("%s: neither PNG nor JPEG", what);
}

// ---------------------------------------------------------------------------------------------- the importer
struct M4 { double m[4][4]; };
M4 identity() { M4 r{}; for(int i = 0; i < 4; ++i) r.m[i][i] = 1; return r; }
M4 mul(const M4& a, const M4& b)
{
  M4 r{};
  for(int i = 0; i < 4; ++i)
    for(int j = 0; j < 4; ++j)
    {
      double s = 0;
      for(int k = 0; k < 4; ++k)
        s += a.m[i][k] * b.m[k][j];
      r.m[i][j] = s;
    }
  return r;
}
void xform(const M4& a, const double v[4], double out[3])
{
  for(int i = 0; i < 3; ++i)
    out[i] = a.m[i][0] * v[0] + a.m[i][1] * v[1] + a.m[i][2] * v[2] + a.m[i][3] * v[3];
}

}  // namespace

struct pt_GltfScene {
  std::vector<pt_VertexAttributes>  vertices;
  std::vector<uint32_t>             indices;
  std::vector<pt_PrimMesh>          primMeshes;
  std::vector<pt_Node>              nodes;
  std::vector<pt_GltfShadeMaterial> materials;
  std::vector<pt_Light>             lights;
  std::vector<Image>                images;    // one per TEXTURE (a texture = sampler + image)
  std::vector<pt_TextureDesc>       textures;
  pt_SceneDesc                      desc{};
  float                             eye[3] = {0, 0, 3}, center[3] = {0, 0, 0}, up[3] = {0, 1, 0}, fov = 60.0f;
  float                             bboxMin[3] = {0, 0, 0}, bboxMax[3] = {0, 0, 0};
};

namespace {

struct Importer {
  JVal               doc;
  std::vector<Bytes> buffers;
  std::string        base;
  pt_GltfScene&      out;
  // raw attributes, concatenated per prim-mesh
  std::vector<float> pos, nrm, tan, uv, col;
  std::map<std::tuple<int, int, int, int, int, int, int>, int> primCache;
  std::vector<std::pair<M4, const JVal*>>                     cameras;
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};

  explicit Importer(pt_GltfScene& o) : out(o) {}

  void load_container(const std::string& path)
  {
    size_t slash = path.find_last_of("/\\");
    base         = slash == std::string::npos ? "" : path.substr(0, slash);
    Bytes raw    = read_file(path);
    Bytes glbBin;
    bool  haveBin = false;
    if(raw.size() >= 12 && !memcmp(raw.data(), "glTF", 4))
    {
      uint32_t version, length;
      memcpy(&version, &raw[4], 4);
      memcpy(&length, &raw[8], 4);
      if(version != 2)
        fail("GLB version %u is not supported", version);
      size_t off = 12;
      bool   haveJson = false;
      while(off + 8 <= std::min<size_t>(length, raw.size()))
      {
        uint32_t clen, ctype;
        memcpy(&clen, &raw[off], 4);
        memcpy(&ctype, &raw[off + 4], 4);
        if(off + 8 + clen > raw.size())
          fail("GLB: truncated chunk");
        if(ctype == 0x4E4F534Au)
        {
          JParser jp{(const char*)&raw[off + 8], (const char*)&raw[off + 8] + clen};
          doc      = jp.parse();
          haveJson = true;
        }
        else if(ctype == 0x004E4942u && !haveBin)
        {
          glbBin.assign(raw.begin() + long(off + 8), raw.begin() + long(off + 8 + clen));
          haveBin = true;
        }
        off += 8 + clen + ((4 - clen % 4) % 4);
      }
      if(!haveJson)
        fail("GLB without a JSON chunk");
    }
    else
    {
      JParser jp{(const char*)raw.data(), (const char*)raw.data() + raw.size()};
      doc = jp.parse();
    }
    const JVal* asset = doc.get("asset");
    std::string ver   = asset ? asset->string("version", "2.0") : "2.0";
    if(ver.empty() || ver[0] != '2')
      fail("only glTF 2.x is supported");
    int i = 0;
    for(const JVal& b : doc.array("buffers"))
    {
      if(b.has("uri"))
        buffers.push_back(read_uri(b.string("uri", ""), base));
      else
      {
        if(!haveBin || i != 0)
          fail("buffer %d has no uri and there is no GLB binary chunk", i);
        buffers.push_back(glbBin);
      }
      ++i;
    }
  }

  // a JSON number that must be a byte offset / length / element count: finite, non-negative, integral, below 2^53
  size_t size_field(const JVal& v, const char* key, const char* what)
  {
    const double d = v.number(key, 0);
    if(!(d >= 0.0) || !(d <= 9007199254740992.0) || d != std::floor(d))
      fail("%s: %s is not a valid size", what, key);
    return size_t(d);
  }

  // (pointer, length, stride) of a buffer view
  void view(int index, const uint8_t*& p, size_t& len, size_t& stride)
  {
    const auto& views = doc.array("bufferViews");
    if(index < 0 || size_t(index) >= views.size())
      fail("bufferView %d out of range", index);
    const JVal& v = views[size_t(index)];
    const int   b = v.integer("buffer", 0);
    if(b < 0 || size_t(b) >= buffers.size())
      fail("buffer %d out of range", b);
    const size_t off  = size_field(v, "byteOffset", "bufferView");
    const size_t size = buffers[size_t(b)].size();
    len               = size_field(v, "byteLength", "bufferView");
    if(off > size || len > size - off)  // written so that it cannot wrap
      fail("bufferView %d exceeds its buffer", index);
    p      = buffers[size_t(b)].data() + off;
    stride = size_field(v, "byteStride", "bufferView");
    if(stride > 252)  // glTF 2.0: byteStride in [4, 252]
      fail("bufferView %d: byteStride %zu out of range", index, stride);
  }

  // accessor as floats (normalised integers converted per the specification) or as raw integers
  void accessor(int index, std::vector<float>* f, std::vector<uint32_t>* u, int& ncomp, size_t& count)
  {
    const auto& accs = doc.array("accessors");
    if(index < 0 || size_t(index) >= accs.size())
      fail("accessor %d out of range", index);
    const JVal& a = accs[size_t(index)];
    const int         ct = a.integer("componentType", 5126);
    const std::string ty = a.string("type", "SCALAR");
    ncomp                = ty == "SCALAR" ? 1 : ty == "VEC2" ? 2 : ty == "VEC3" ? 3 : ty == "VEC4" ? 4 : ty == "MAT2" ? 4 : ty == "MAT3" ? 9 : 16;
    count                = size_field(a, "count", "accessor");
    if(count > (size_t(1) << 31))
      fail("accessor %d: count too large", index);
    const size_t csize   = (ct == 5120 || ct == 5121) ? 1 : (ct == 5122 || ct == 5123) ? 2 : 4;
    const size_t elem    = csize * size_t(ncomp);
    const bool   norm    = a.boolean("normalized", false);
    const uint8_t* p = nullptr;
    size_t         len = 0, stride = 0;
    static const uint8_t zeros[64] = {0};
    const bool           hasView   = a.has("bufferView");
    if(hasView)
    {
      view(a.integer("bufferView", -1), p, len, stride);
      const size_t off = size_field(a, "byteOffset", "accessor");
      if(stride == 0)
        stride = elem;
      if(stride < elem)
        fail("accessor %d: byteStride smaller than an element", index);
      // count elements of `elem` bytes, `stride` apart, starting at `off`, inside `len` bytes -- without overflow
      if(count && (off > len || elem > len - off || (count - 1) > (len - off - elem) / stride))
        fail("accessor %d exceeds its bufferView", index);
      p += off;
    }
    if(f) f->resize(count * size_t(ncomp));
    if(u) u->resize(count * size_t(ncomp));
    for(size_t i = 0; i < count; ++i)
    {
      const uint8_t* e = hasView ? p + stride * i : zeros;
      for(int k = 0; k < ncomp; ++k)
      {
        const uint8_t* c = e + csize * size_t(k);
        double         v;
        uint32_t       iv = 0;
        switch(ct)
        {
          case 5120: { int8_t x; memcpy(&x, c, 1); v = x; iv = uint32_t(x); if(norm) v = std::max(double(float(x) / 127.0f), -1.0); break; }
          case 5121: { uint8_t x = *c; v = x; iv = x; if(norm) v = double(float(x) / 255.0f); break; }
          case 5122: { int16_t x; memcpy(&x, c, 2); v = x; iv = uint32_t(x); if(norm) v = std::max(double(float(x) / 32767.0f), -1.0); break; }
          case 5123: { uint16_t x; memcpy(&x, c, 2); v = x; iv = x; if(norm) v = double(float(x) / 65535.0f); break; }
          case 5125: { uint32_t x; memcpy(&x, c, 4); v = x; iv = x; break; }
          default: { float x; memcpy(&x, c, 4); v = x; iv = uint32_t(x); break; }
        }
        if(f) (*f)[i * size_t(ncomp) + size_t(k)] = float(v);
        if(u) (*u)[i * size_t(ncomp) + size_t(k)] = iv;
      }
    }
    if(const JVal* sp = a.get("sparse"))
    {  // substituted elements on top of the (possibly absent = zero) base data
      const size_t n  = size_field(*sp, "count", "sparse accessor");
      const JVal*  si = sp->get("indices");
      const JVal*  sv = sp->get("values");
      if(!si || !sv)
        fail("accessor %d: incomplete sparse block", index);
      const uint8_t *ip, *vp;
      size_t         il, vl, st;
      view(si->integer("bufferView", -1), ip, il, st);
      view(sv->integer("bufferView", -1), vp, vl, st);
      const size_t ioff = size_field(*si, "byteOffset", "sparse indices"), voff = size_field(*sv, "byteOffset", "sparse values");
      const int    ict  = si->integer("componentType", 5125);
      const size_t isz  = ict == 5121 ? 1 : ict == 5123 ? 2 : 4;
      if(n > count || ioff > il || n > (il - ioff) / isz || voff > vl || n > (vl - voff) / elem)
        fail("accessor %d: sparse data exceeds its bufferView", index);
      long long prev = -1;
      for(size_t j = 0; j < n; ++j)
      {
        uint32_t ix = 0;
        memcpy(&ix, ip + ioff + isz * j, isz);
        if((long long)ix <= prev || ix >= count)
          fail("sparse accessor indices must be strictly increasing and below count");
        prev = ix;
        const uint8_t* e = vp + voff + elem * j;
        for(int k = 0; k < ncomp; ++k)
        {
          const uint8_t* c = e + csize * size_t(k);
          double         v;
          uint32_t       iv = 0;
          switch(ct)
          {
            case 5120: { int8_t x; memcpy(&x, c, 1); v = x; iv = uint32_t(x); if(norm) v = std::max(double(float(x) / 127.0f), -1.0); break; }
            case 5121: { uint8_t x = *c; v = x; iv = x; if(norm) v = double(float(x) / 255.0f); break; }
            case 5122: { int16_t x; memcpy(&x, c, 2); v = x; iv = uint32_t(x); if(norm) v = std::max(double(float(x) / 32767.0f), -1.0); break; }
            case 5123: { uint16_t x; memcpy(&x, c, 2); v = x; iv = x; if(norm) v = double(float(x) / 65535.0f); break; }
            case 5125: { uint32_t x; memcpy(&x, c, 4); v = x; iv = x; break; }
            default: { float x; memcpy(&x, c, 4); v = x; iv = uint32_t(x); break; }
          }
          if(f) (*f)[size_t(ix) * size_t(ncomp) + size_t(k)] = float(v);
          if(u) (*u)[size_t(ix) * size_t(ncomp) + size_t(k)] = iv;
        }
      }
    }
  }

  static int tex_index(const JVal* info) { return (info && info->t == JVal::Obj) ? info->integer("index", -1) : -1; }

  pt_GltfShadeMaterial default_material()
  {
    pt_GltfShadeMaterial m;
    memset(&m, 0, sizeof(m));
    for(int i = 0; i < 4; ++i) m.pbrBaseColorFactor[i] = 1.0f;
    m.pbrBaseColorTexture = -1; m.pbrMetallicFactor = 1.0f; m.pbrRoughnessFactor = 1.0f; m.pbrMetallicRoughnessTexture = -1;
    m.emissiveTexture = -1; m.alphaMode = PT_ALPHA_OPAQUE; m.alphaCutoff = 0.5f; m.normalTexture = -1; m.normalTextureScale = 1.0f;
    for(int i = 0; i < 4; ++i) m.uvTransform[i * 5] = 1.0f;
    m.transmissionTexture = -1; m.ior = 1.5f; m.anisotropyDirection[1] = 1.0f;
    for(int i = 0; i < 3; ++i) m.attenuationColor[i] = 1.0f;
    m.thicknessTexture = -1; m.attenuationDistance = 3.4028235e38f; m.clearcoatTexture = -1; m.clearcoatRoughnessTexture = -1;
    return m;
  }

  pt_GltfShadeMaterial material(const JVal& g)
  {
    pt_GltfShadeMaterial m = default_material();
    static const JVal    none;
    const JVal&          pbr = g.get("pbrMetallicRoughness") ? *g.get("pbrMetallicRoughness") : none;
    auto vecN = [](const JVal* a, float* dst, int n) {
      if(a && a->t == JVal::Arr)
        for(int i = 0; i < n && size_t(i) < a->arr.size(); ++i)
          dst[i] = float(a->arr[size_t(i)].num);
    };
    vecN(pbr.get("baseColorFactor"), m.pbrBaseColorFactor, 4);
    m.pbrBaseColorTexture         = tex_index(pbr.get("baseColorTexture"));
    m.pbrMetallicFactor           = float(pbr.number("metallicFactor", 1.0));
    m.pbrRoughnessFactor          = float(pbr.number("roughnessFactor", 1.0));
    m.pbrMetallicRoughnessTexture = tex_index(pbr.get("metallicRoughnessTexture"));
    m.emissiveTexture             = tex_index(g.get("emissiveTexture"));
    vecN(g.get("emissiveFactor"), m.emissiveFactor, 3);
    const std::string am = g.string("alphaMode", "OPAQUE");
    m.alphaMode          = am == "MASK" ? PT_ALPHA_MASK : am == "BLEND" ? PT_ALPHA_BLEND : PT_ALPHA_OPAQUE;
    m.alphaCutoff        = float(g.number("alphaCutoff", 0.5));
    m.doubleSided        = g.boolean("doubleSided", false) ? 1 : 0;
    m.normalTexture      = tex_index(g.get("normalTexture"));
    m.normalTextureScale = g.get("normalTexture") ? float(g.get("normalTexture")->number("scale", 1.0)) : 1.0f;
    // KHR_texture_transform of the base-colour texture, in the row-vector convention the shader applies:
    // (u, v, 1, 1) * M with uv' = T * R * S * uv (gltf_material.glsl:115)
    if(const JVal* bct = pbr.get("baseColorTexture"))
      if(const JVal* ex = bct->get("extensions"))
        if(const JVal* tt = ex->get("KHR_texture_transform"))
        {
          double off[2] = {0, 0}, sc[2] = {1, 1};
          if(const JVal* o = tt->get("offset")) for(int i = 0; i < 2 && size_t(i) < o->arr.size(); ++i) off[i] = o->arr[size_t(i)].num;
          if(const JVal* s = tt->get("scale")) for(int i = 0; i < 2 && size_t(i) < s->arr.size(); ++i) sc[i] = s->arr[size_t(i)].num;
          const double r = tt->number("rotation", 0.0), c = std::cos(r), s = std::sin(r);
          m.uvTransform[0] = float(c * sc[0]);  m.uvTransform[1] = float(s * sc[1]); m.uvTransform[2] = float(off[0]);
          m.uvTransform[4] = float(-s * sc[0]); m.uvTransform[5] = float(c * sc[1]); m.uvTransform[6] = float(off[1]);
        }
    const JVal& ext = g.get("extensions") ? *g.get("extensions") : none;
    m.unlit         = ext.has("KHR_materials_unlit") ? 1 : 0;
    if(const JVal* t = ext.get("KHR_materials_transmission"))
    {
      m.transmissionFactor  = float(t->number("transmissionFactor", 0.0));
      m.transmissionTexture = tex_index(t->get("transmissionTexture"));
    }
    if(const JVal* t = ext.get("KHR_materials_ior"))
      m.ior = float(t->number("ior", 1.5));
    if(const JVal* t = ext.get("KHR_materials_anisotropy"))
    {
      const double rot         = t->number("anisotropyRotation", 0.0);
      m.anisotropy             = float(t->number("anisotropyStrength", 0.0));
      m.anisotropyDirection[0] = float(std::sin(rot));  // src/scene.cpp:366
      m.anisotropyDirection[1] = float(std::cos(rot));
      m.anisotropyDirection[2] = 0.0f;
    }
    if(const JVal* t = ext.get("KHR_materials_volume"))
    {
      vecN(t->get("attenuationColor"), m.attenuationColor, 3);
      m.thicknessFactor     = float(t->number("thicknessFactor", 0.0));
      m.thicknessTexture    = tex_index(t->get("thicknessTexture"));
      m.attenuationDistance = float(std::min(t->number("attenuationDistance", 3.4028235e38), 3.4028235e38));
    }
    if(const JVal* t = ext.get("KHR_materials_clearcoat"))
    {
      m.clearcoatFactor           = float(t->number("clearcoatFactor", 0.0));
      m.clearcoatRoughness        = float(t->number("clearcoatRoughnessFactor", 0.0));
      m.clearcoatTexture          = tex_index(t->get("clearcoatTexture"));
      m.clearcoatRoughnessTexture = tex_index(t->get("clearcoatRoughnessTexture"));
    }
    if(const JVal* t = ext.get("KHR_materials_sheen"))
    {
      float c4[4] = {0, 0, 0, float(t->number("sheenRoughnessFactor", 0.0))};
      vecN(t->get("sheenColorFactor"), c4, 3);
      uint32_t packed = 0;
      for(int i = 0; i < 4; ++i)  // glm::packUnorm4x8 (src/scene.cpp:376)
        packed |= uint32_t(std::nearbyintf(std::fmin(std::fmax(c4[i], 0.0f), 1.0f) * 255.0f)) << (8 * i);
      m.sheen = packed;
    }
    return m;
  }

  M4 local_matrix(const JVal& n)
  {
    M4 r = identity();
    if(const JVal* mm = n.get("matrix"))
    {
      if(mm->arr.size() == 16)
        for(int c = 0; c < 4; ++c)
          for(int rr = 0; rr < 4; ++rr)
            r.m[rr][c] = mm->arr[size_t(c * 4 + rr)].num;  // column-major in the file
      return r;
    }
    double t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
    auto   rd   = [&](const char* k, double* d, int cnt) { if(const JVal* a = n.get(k)) for(int i = 0; i < cnt && size_t(i) < a->arr.size(); ++i) d[i] = a->arr[size_t(i)].num; };
    rd("translation", t, 3); rd("rotation", q, 4); rd("scale", s, 3);
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
                            {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
                            {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
    for(int i = 0; i < 3; ++i)
    {
      for(int j = 0; j < 3; ++j)
        r.m[i][j] = R[i][j] * s[j];
      r.m[i][3] = t[i];
    }
    return r;
  }

  // ---- attribute synthesis (files without NORMAL / TANGENT; see the header comment)
  static void synth_normals(const std::vector<float>& P, const std::vector<uint32_t>& idx, std::vector<float>& N)
  {
    const size_t        n = P.size() / 3;
    std::vector<double> acc(n * 3, 0.0);
    for(size_t t = 0; t + 2 < idx.size(); t += 3)
    {
      const uint32_t a = idx[t], b = idx[t + 1], c = idx[t + 2];
      // (float32 edge vectors and cross product like the Python importer, accumulated in double)
      const float e1[3] = {P[b * 3] - P[a * 3], P[b * 3 + 1] - P[a * 3 + 1], P[b * 3 + 2] - P[a * 3 + 2]};
      const float e2[3] = {P[c * 3] - P[a * 3], P[c * 3 + 1] - P[a * 3 + 1], P[c * 3 + 2] - P[a * 3 + 2]};
      const float fn[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
      for(uint32_t v : {a, b, c})
        for(int k = 0; k < 3; ++k)
          acc[v * 3 + size_t(k)] += double(fn[k]);
    }
    N.resize(n * 3);
    for(size_t v = 0; v < n; ++v)
    {
      const double l = std::sqrt(acc[v * 3] * acc[v * 3] + acc[v * 3 + 1] * acc[v * 3 + 1] + acc[v * 3 + 2] * acc[v * 3 + 2]);
      for(int k = 0; k < 3; ++k)
        N[v * 3 + size_t(k)] = l > 0 ? float(acc[v * 3 + size_t(k)] / std::max(l, 1e-300)) : (k == 2 ? 1.0f : 0.0f);
    }
  }
  static void default_tangent(const float* n, float* t4)
  {
    const float ref[3] = {std::fabs(n[1]) < 0.99f ? 0.f : 1.f, std::fabs(n[1]) < 0.99f ? 1.f : 0.f, 0.f};
    float       t[3]   = {ref[1] * n[2] - ref[2] * n[1], ref[2] * n[0] - ref[0] * n[2], ref[0] * n[1] - ref[1] * n[0]};
    const float l      = std::max(std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]), 1e-20f);
    t4[0] = t[0] / l; t4[1] = t[1] / l; t4[2] = t[2] / l; t4[3] = 1.0f;
  }
  static void synth_tangents(const std::vector<float>& P, const std::vector<float>& N, const std::vector<float>& UV, const std::vector<uint32_t>& idx, std::vector<float>& T)
  {
    const size_t        n = P.size() / 3;
    std::vector<double> tan(n * 3, 0.0), bit(n * 3, 0.0);
    for(size_t t = 0; t + 2 < idx.size(); t += 3)
    {
      const uint32_t i0 = idx[t], i1 = idx[t + 1], i2 = idx[t + 2];
      double         e1[3], e2[3];
      for(int k = 0; k < 3; ++k)
      {
        e1[k] = double(P[i1 * 3 + size_t(k)]) - double(P[i0 * 3 + size_t(k)]);
        e2[k] = double(P[i2 * 3 + size_t(k)]) - double(P[i0 * 3 + size_t(k)]);
      }
      const double d1[2] = {double(UV[i1 * 2]) - double(UV[i0 * 2]), double(UV[i1 * 2 + 1]) - double(UV[i0 * 2 + 1])};
      const double d2[2] = {double(UV[i2 * 2]) - double(UV[i0 * 2]), double(UV[i2 * 2 + 1]) - double(UV[i0 * 2 + 1])};
      const double det   = d1[0] * d2[1] - d2[0] * d1[1];
      const double r     = std::fabs(det) > 1e-20 ? 1.0 / det : 0.0;
      for(uint32_t v : {i0, i1, i2})
        for(int k = 0; k < 3; ++k)
        {
          tan[v * 3 + size_t(k)] += (e1[k] * d2[1] - e2[k] * d1[1]) * r;
          bit[v * 3 + size_t(k)] += (e2[k] * d1[0] - e1[k] * d2[0]) * r;
        }
    }
    T.resize(n * 4);
    for(size_t v = 0; v < n; ++v)
    {
      const double nn[3] = {N[v * 3], N[v * 3 + 1], N[v * 3 + 2]};
      const double dt    = nn[0] * tan[v * 3] + nn[1] * tan[v * 3 + 1] + nn[2] * tan[v * 3 + 2];
      double       t[3]  = {tan[v * 3] - nn[0] * dt, tan[v * 3 + 1] - nn[1] * dt, tan[v * 3 + 2] - nn[2] * dt};
      const double l     = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
      if(l > 1e-12)
      {
        for(int k = 0; k < 3; ++k) t[k] /= std::max(l, 1e-300);
        const double cx[3] = {nn[1] * t[2] - nn[2] * t[1], nn[2] * t[0] - nn[0] * t[2], nn[0] * t[1] - nn[1] * t[0]};
        const double h     = (cx[0] * bit[v * 3] + cx[1] * bit[v * 3 + 1] + cx[2] * bit[v * 3 + 2]) < 0.0 ? -1.0 : 1.0;
        T[v * 4] = float(t[0]); T[v * 4 + 1] = float(t[1]); T[v * 4 + 2] = float(t[2]); T[v * 4 + 3] = float(h);
      }
      else
        default_tangent(&N[v * 3], &T[v * 4]);
    }
  }

  int prim_mesh(const JVal& p)
  {
    if(p.integer("mode", 4) != 4)
      return -1;  // only triangle lists reach the BLAS builder
    static const JVal none;
    const JVal&       at = p.get("attributes") ? *p.get("attributes") : none;
    if(!at.has("POSITION"))
      return -1;
    const int nmat = int(out.materials.size());
    auto key = std::make_tuple(at.integer("POSITION", -1), at.integer("NORMAL", -1), at.integer("TEXCOORD_0", -1), at.integer("TANGENT", -1), at.integer("COLOR_0", -1),
                               p.integer("indices", -1), p.integer("material", -1));
    auto it = primCache.find(key);
    if(it != primCache.end())
      return it->second;
    std::vector<float>    P, N, UV, T, C;
    std::vector<uint32_t> idx;
    int                   nc;
    size_t                n, cnt;
    accessor(at.integer("POSITION", -1), &P, nullptr, nc, n);
    if(nc != 3)
      fail("POSITION must be VEC3");
    if(p.has("indices"))
    {
      accessor(p.integer("indices", -1), nullptr, &idx, nc, cnt);
    }
    else
    {
      idx.resize(n);
      for(size_t i = 0; i < n; ++i) idx[i] = uint32_t(i);
    }
    idx.resize(idx.size() / 3 * 3);
    for(uint32_t i : idx)
      if(i >= n)
        fail("primitive index out of range");
    auto take = [&](const char* name, std::vector<float>& dst, int want, float fill) -> bool {
      if(!at.has(name))
        return false;
      std::vector<float> raw;
      int                c;
      size_t             k;
      accessor(at.integer(name, -1), &raw, nullptr, c, k);
      if(k != n)
        fail("%s has %zu elements, POSITION has %zu", name, k, n);
      dst.assign(n * size_t(want), fill);
      for(size_t i = 0; i < n; ++i)
        for(int j = 0; j < std::min(c, want); ++j)
          dst[i * size_t(want) + size_t(j)] = raw[i * size_t(c) + size_t(j)];
      return true;
    };
    if(!take("NORMAL", N, 3, 0.f)) synth_normals(P, idx, N);
    if(!take("TEXCOORD_0", UV, 2, 0.f)) UV.assign(n * 2, 0.f);
    if(!take("TANGENT", T, 4, 1.f))
    {
      if(!idx.empty())
        synth_tangents(P, N, UV, idx, T);
      else
      {
        T.resize(n * 4);
        for(size_t v = 0; v < n; ++v) default_tangent(&N[v * 3], &T[v * 4]);
      }
    }
    if(!take("COLOR_0", C, 4, 1.f)) C.assign(n * 4, 1.f);
    int mat = p.integer("material", -1);
    if(mat < 0 || mat >= nmat)
      mat = 0;
    pt_PrimMesh pm;
    pm.vertexOffset  = uint32_t(pos.size() / 3);
    pm.vertexCount   = uint32_t(n);
    pm.firstIndex    = uint32_t(out.indices.size());
    pm.indexCount    = uint32_t(idx.size());
    pm.materialIndex = mat;
    pos.insert(pos.end(), P.begin(), P.end()); nrm.insert(nrm.end(), N.begin(), N.end()); uv.insert(uv.end(), UV.begin(), UV.end());
    tan.insert(tan.end(), T.begin(), T.end()); col.insert(col.end(), C.begin(), C.end());
    out.indices.insert(out.indices.end(), idx.begin(), idx.end());
    out.primMeshes.push_back(pm);
    const int id = int(out.primMeshes.size()) - 1;
    primCache[key] = id;
    return id;
  }

  std::vector<uint8_t> onPath;  // nodes on the current root-to-node path: a child that is already on it closes a cycle
  void visit(int ni, const M4& parent)
  {
    const auto& nodes = doc.array("nodes");
    if(ni < 0 || size_t(ni) >= nodes.size())
      fail("node %d out of range", ni);
    if(onPath.size() != nodes.size())
      onPath.assign(nodes.size(), 0);
    if(onPath[size_t(ni)])
      fail("node hierarchy has a cycle through node %d", ni);
    struct Mark {
      uint8_t& m;
      explicit Mark(uint8_t& r) : m(r) { m = 1; }
      ~Mark() { m = 0; }
    } mark(onPath[size_t(ni)]);
    const JVal& n     = nodes[size_t(ni)];
    const M4    world = mul(parent, local_matrix(n));
    if(n.has("mesh"))
    {
      const auto& meshes = doc.array("meshes");
      const int   mi     = n.integer("mesh", -1);
      if(mi < 0 || size_t(mi) >= meshes.size())
        fail("mesh %d out of range", mi);
      for(const JVal& p : meshes[size_t(mi)].array("primitives"))
      {
        const int pm = prim_mesh(p);
        if(pm < 0)
          continue;
        pt_Node nd;
        for(int c = 0; c < 4; ++c)
          for(int r = 0; r < 4; ++r)
            nd.worldMatrix[c * 4 + r] = float(world.m[r][c]);
        nd.primMesh = pm;
        out.nodes.push_back(nd);
        const pt_PrimMesh& m = out.primMeshes[size_t(pm)];
        float              bl[3] = {3.0e38f, 3.0e38f, 3.0e38f}, bh[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for(uint32_t v = 0; v < m.vertexCount; ++v)
          for(int k = 0; k < 3; ++k)
          {
            bl[k] = std::min(bl[k], pos[(m.vertexOffset + v) * 3 + size_t(k)]);
            bh[k] = std::max(bh[k], pos[(m.vertexOffset + v) * 3 + size_t(k)]);
          }
        for(int c = 0; c < 8; ++c)
        {
          const double v[4] = {(c & 4) ? bh[0] : bl[0], (c & 2) ? bh[1] : bl[1], (c & 1) ? bh[2] : bl[2], 1.0};
          double       w[3];
          xform(world, v, w);
          for(int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], w[k]); hi[k] = std::max(hi[k], w[k]); }
        }
      }
    }
    if(n.has("camera"))
    {
      const auto& cams = doc.array("cameras");
      const int   ci   = n.integer("camera", -1);
      if(ci >= 0 && size_t(ci) < cams.size() && cams[size_t(ci)].string("type", "perspective") == "perspective")
        cameras.emplace_back(world, &cams[size_t(ci)]);
    }
    if(const JVal* ex = n.get("extensions"))
      if(const JVal* le = ex->get("KHR_lights_punctual"))
      {
        const JVal* de = doc.get("extensions");
        const JVal* lp = de ? de->get("KHR_lights_punctual") : nullptr;
        const int   li = le->integer("light", -1);
        if(lp && li >= 0 && size_t(li) < lp->array("lights").size())
        {
          const JVal& L = lp->array("lights")[size_t(li)];
          pt_Light    l;
          memset(&l, 0, sizeof(l));
          const double o4[4] = {0, 0, 0, 1}, d4[4] = {0, 0, -1, 0};
          double       v[3];
          xform(world, o4, v); for(int k = 0; k < 3; ++k) l.position[k] = float(v[k]);   // src/scene.cpp:309-310
          xform(world, d4, v); for(int k = 0; k < 3; ++k) l.direction[k] = float(v[k]);
          l.color[0] = l.color[1] = l.color[2] = 1.0f;
          if(const JVal* c = L.get("color")) for(int k = 0; k < 3 && size_t(k) < c->arr.size(); ++k) l.color[k] = float(c->arr[size_t(k)].num);
          static const JVal none;
          const JVal&       spot = L.get("spot") ? *L.get("spot") : none;
          l.innerConeCos = float(std::cos(spot.number("innerConeAngle", 0.0)));
          l.outerConeCos = float(std::cos(spot.number("outerConeAngle", M_PI / 4)));
          l.range        = float(L.number("range", 0.0));
          l.intensity    = float(L.number("intensity", 1.0));
          const std::string ty = L.string("type", "point");
          l.type = ty == "directional" ? 0 : ty == "spot" ? 2 : 1;  // LightType_Directional / _Point / _Spot (host_device.h:211-213)
          out.lights.push_back(l);
        }
      }
    for(const JVal& c : n.array("children"))
      visit(int(c.num), world);
  }

  void run(const std::string& path)
  {
    load_container(path);
    // textures = (sampler, image) pairs; materials index TEXTURES (src/scene.cpp:552-575)
    const auto&                         images = doc.array("images");
    std::map<int, std::shared_ptr<Image>> decoded;
    for(const JVal& t : doc.array("textures"))
    {
      const int src = t.integer("source", -1);
      Image     im;
      pt_TextureDesc td;
      memset(&td, 0, sizeof(td));
      if(src < 0 || size_t(src) >= images.size())
      {  // "Incorrect source image" -> 1x1 white dummy with a default-constructed sampler (src/scene.cpp:554-559)
        im.w = im.h = 1;
        im.rgba.assign(4, 255);
        pt_sampler_from_gltf(0, 0, 0, 0, 0, &td);
      }
      else
      {
        if(!decoded.count(src))
        {
          const JVal& ji = images[size_t(src)];
          Bytes       data;
          if(ji.has("bufferView"))
          {
            const uint8_t* p;
            size_t         len, stride;
            view(ji.integer("bufferView", -1), p, len, stride);
            data.assign(p, p + len);
          }
          else if(ji.has("uri"))
            data = read_uri(ji.string("uri", ""), base);
          else
            fail("image %d has neither uri nor bufferView", src);
          char what[32];
          snprintf(what, sizeof(what), "image %d", src);
          decoded[src] = std::make_shared<Image>(decode_image(data, what));
        }
        im = *decoded[src];
        const int s = t.integer("sampler", -1);
        if(s < 0 || size_t(s) >= doc.array("samplers").size())
          pt_sampler_from_gltf(0, 0, 0, 0, 0, &td);
        else
        {  // tinygltf defaults: filters -1 (-> enum 0 = NEAREST through the reference's std::map lookup), wrap REPEAT
          const JVal& sm = doc.array("samplers")[size_t(s)];
          pt_sampler_from_gltf(1, sm.integer("magFilter", -1), sm.integer("minFilter", -1), sm.integer("wrapS", 10497), sm.integer("wrapT", 10497), &td);
        }
      }
      td.width = im.w; td.height = im.h;
      out.images.push_back(std::move(im));
      out.textures.push_back(td);
    }
    for(size_t i = 0; i < out.textures.size(); ++i)
      out.textures[i].rgba8 = out.images[i].rgba.data();

    for(const JVal& g : doc.array("materials"))
      out.materials.push_back(material(g));
    if(out.materials.empty())
      out.materials.push_back(default_material());
    const int ntex = int(out.textures.size());
    for(const auto& m : out.materials)
      for(int id : {m.pbrBaseColorTexture, m.pbrMetallicRoughnessTexture, m.emissiveTexture, m.normalTexture, m.transmissionTexture, m.thicknessTexture, m.clearcoatTexture,
                    m.clearcoatRoughnessTexture})
        if(id >= ntex)
          fail("material references texture %d of %d", id, ntex);

    const auto& scenes = doc.array("scenes");
    if(!scenes.empty())
    {
      const int si = doc.integer("scene", 0);
      if(si < 0 || size_t(si) >= scenes.size())
        fail("scene %d out of range", si);
      for(const JVal& r : scenes[size_t(si)].array("nodes"))
        visit(int(r.num), identity());
    }
    else
      for(size_t i = 0; i < doc.array("nodes").size(); ++i)
        visit(int(i), identity());

    out.vertices.resize(pos.size() / 3);
    if(!out.vertices.empty() && pt_pack_vertices(uint32_t(out.vertices.size()), pos.data(), nrm.data(), tan.data(), uv.data(), col.data(), out.vertices.data()) != PT_OK)
      fail("pt_pack_vertices failed");

    const bool haveBox = lo[0] <= hi[0];
    for(int k = 0; k < 3; ++k) { out.bboxMin[k] = haveBox ? float(lo[k]) : 0.f; out.bboxMax[k] = haveBox ? float(hi[k]) : 0.f; }
    const double diag = haveBox ? std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2])) : 1.0;
    if(!cameras.empty())
    {  // the first camera of the scene (src/scene.cpp:284-288)
      const M4&    w    = cameras[0].first;
      const JVal&  cam  = *cameras[0].second;
      const double o4[4] = {0, 0, 0, 1}, f4[4] = {0, 0, -1, 0}, u4[4] = {0, 1, 0, 0};
      double       e[3], f[3], u[3];
      xform(w, o4, e); xform(w, f4, f); xform(w, u4, u);
      const double fl = std::max(std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]), 1e-30);
      double       focus = std::max(diag * 0.5, 1e-3);
      if(const JVal* ex = cam.get("extras"))
        if(ex->number("pt_focus_distance", 0.0) > 0.0)
          focus = ex->number("pt_focus_distance", 0.0);
      static const JVal none;
      const JVal&       persp = cam.get("perspective") ? *cam.get("perspective") : none;
      for(int k = 0; k < 3; ++k)
      {
        out.eye[k]    = float(e[k]);
        out.center[k] = float(e[k] + f[k] / fl * focus);
        out.up[k]     = float(u[k]);
      }
      out.fov = float(persp.number("yfov", 60.0 * M_PI / 180.0) * 180.0 / M_PI);
    }
    else if(haveBox)
    {  // no camera in the file: look at the bounding box from +z so that it fits the default 60 degree frustum
      const double c[3] = {(lo[0] + hi[0]) * 0.5, (lo[1] + hi[1]) * 0.5, (lo[2] + hi[2]) * 0.5}, r = diag * 0.5;
      for(int k = 0; k < 3; ++k) { out.center[k] = float(c[k]); out.eye[k] = float(c[k]); out.up[k] = k == 1 ? 1.f : 0.f; }
      out.eye[2] = float(c[2] + r / std::sin(30.0 * M_PI / 180.0));
      out.fov    = 60.0f;
    }
    pt_SceneDesc& d = out.desc;
    d.vertices = out.vertices.data(); d.numVertices = uint32_t(out.vertices.size());
    d.indices = out.indices.data(); d.numIndices = uint32_t(out.indices.size());
    d.primMeshes = out.primMeshes.data(); d.numPrimMeshes = uint32_t(out.primMeshes.size());
    d.nodes = out.nodes.data(); d.numNodes = uint32_t(out.nodes.size());
    d.materials = out.materials.data(); d.numMaterials = uint32_t(out.materials.size());
    d.lights = out.lights.empty() ? nullptr : out.lights.data(); d.numLights = uint32_t(out.lights.size());
    d.textures = out.textures.empty() ? nullptr : out.textures.data(); d.numTextures = uint32_t(out.textures.size());
  }
};

}  // namespace

extern "C" {

int pt_gltf_load(const char* path, pt_GltfScene** out_scene, char* err, size_t err_len)
{
  if(err && err_len)
    err[0] = 0;
  if(!path || !out_scene)
    return PT_ERR_INVALID;
  *out_scene = nullptr;
  std::unique_ptr<pt_GltfScene> sc(new pt_GltfScene());
  try
  {
    Importer imp(*sc);
    imp.run(path);
  }
  catch(const std::exception& e)
  {
    if(err && err_len)
      snprintf(err, err_len, "%s", e.what());
    return PT_ERR_INVALID;
  }
  *out_scene = sc.release();
  return PT_OK;
}

const pt_SceneDesc* pt_gltf_desc(const pt_GltfScene* s) { return s ? &s->desc : nullptr; }

int pt_gltf_camera(const pt_GltfScene* s, float eye[3], float center[3], float up[3], float* fov_degrees)
{
  if(!s)
    return PT_ERR_INVALID;
  for(int k = 0; k < 3; ++k)
  {
    if(eye) eye[k] = s->eye[k];
    if(center) center[k] = s->center[k];
    if(up) up[k] = s->up[k];
  }
  if(fov_degrees)
    *fov_degrees = s->fov;
  return PT_OK;
}

int pt_gltf_bounds(const pt_GltfScene* s, float bbox_min[3], float bbox_max[3])
{
  if(!s)
    return PT_ERR_INVALID;
  for(int k = 0; k < 3; ++k)
  {
    if(bbox_min) bbox_min[k] = s->bboxMin[k];
    if(bbox_max) bbox_max[k] = s->bboxMax[k];
  }
  return PT_OK;
}

void pt_gltf_free(pt_GltfScene* s) { delete s; }

}  // extern "C"
