// Radiance RGBE (.hdr / .pic) reader of libptmi.so -- the decoder behind the reference's environment loader.
//
// The reference calls stbi_loadf(path, &w, &h, &comp, STBI_rgb_alpha) (src/hdr_sampling.cpp:64; stb_image lives in the un-vendored
// nvpro_core, so its algorithm is restated from the Radiance file format): text header starting "#?RADIANCE" or "#?RGBE" with a
// FORMAT=32-bit_rle_rgbe line, terminated by an empty line; resolution line "-Y <height> +X <width>" (top-to-bottom, left-to-right:
// the only layout stb_image accepts); scanlines either flat RGBE quadruples or "new" run-length encoding (2, 2, width-hi, width-lo,
// then the four channels separately: a count byte > 128 repeats the next byte count-128 times, otherwise count literal bytes).
// RGBE -> float like stb_image: value = mantissa * 2^(exponent - 136) per channel, exponent 0 -> 0; alpha = 1.  No vertical flip.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/pt_api.h"

namespace {
struct Fail {
  std::string msg;
};
[[noreturn]] void fail(const char* m) { throw Fail{m}; }

struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  int            get()
  {
    if(p >= end)
      fail("HDR: unexpected end of file");
    return *p++;
  }
  std::string line()
  {
    std::string s;
    while(p < end && *p != '\n')
    {
      if(s.size() > 1024)
        fail("HDR: header line too long");
      s.push_back(char(*p++));
    }
    if(p < end)
      ++p;
    return s;
  }
};

void rgbe_to_float(const uint8_t* in, float* out)
{
  if(in[3] != 0)
  {
    const float f = std::ldexp(1.0f, int(in[3]) - (128 + 8));
    out[0] = float(in[0]) * f;
    out[1] = float(in[1]) * f;
    out[2] = float(in[2]) * f;
  }
  else
    out[0] = out[1] = out[2] = 0.0f;
  out[3] = 1.0f;
}

void decode(const std::vector<uint8_t>& file, std::vector<float>& rgba, int& w, int& h)
{
  Reader r{file.data(), file.data() + file.size()};
  const std::string magic = r.line();
  if(magic != "#?RADIANCE" && magic != "#?RGBE")
    fail("HDR: not a Radiance file (missing #?RADIANCE)");
  bool format = false;
  for(;;)
  {
    if(r.p >= r.end)
      fail("HDR: header not terminated");
    const std::string l = r.line();
    if(l.empty())
      break;
    if(l == "FORMAT=32-bit_rle_rgbe")
      format = true;
  }
  if(!format)
    fail("HDR: unsupported format (only 32-bit_rle_rgbe)");
  const std::string res = r.line();
  long              hh = 0, ww = 0;
  if(std::sscanf(res.c_str(), "-Y %ld +X %ld", &hh, &ww) != 2)
    fail("HDR: unsupported data layout (only -Y <h> +X <w>)");
  if(hh <= 0 || ww <= 0 || hh > 65536 || ww > 65536)
    fail("HDR: bad image size");
  w = int(ww);
  h = int(hh);
  rgba.assign(size_t(w) * h * 4, 0.0f);
  std::vector<uint8_t> scan(size_t(w) * 4);
  auto                 flat_rest = [&](int x0, int y0, const uint8_t first[4]) {
    // the file is not run-length encoded: `first` is pixel (x0, y0), everything after it is flat RGBE
    int x = x0, y = y0;
    uint8_t px[4] = {first[0], first[1], first[2], first[3]};
    for(;;)
    {
      rgbe_to_float(px, &rgba[(size_t(y) * w + x) * 4]);
      if(++x == w)
      {
        x = 0;
        if(++y == h)
          return;
      }
      for(int k = 0; k < 4; ++k)
        px[k] = uint8_t(r.get());
    }
  };
  if(w < 8 || w >= 32768)
  {
    uint8_t px[4];
    for(int k = 0; k < 4; ++k)
      px[k] = uint8_t(r.get());
    flat_rest(0, 0, px);
    return;
  }
  for(int y = 0; y < h; ++y)
  {
    uint8_t c[4];
    for(int k = 0; k < 4; ++k)
      c[k] = uint8_t(r.get());
    if(c[0] != 2 || c[1] != 2 || (c[2] & 0x80))
    {
      if(y != 0)
        fail("HDR: corrupt run-length scanline header");
      flat_rest(0, 0, c);  // a flat file: stb_image takes this exit on the first scanline only
      return;
    }
    if(((int(c[2]) << 8) | c[3]) != w)
      fail("HDR: scanline width does not match the image width");
    for(int k = 0; k < 4; ++k)
    {
      int x = 0;
      while(x < w)
      {
        int count = r.get();
        if(count > 128)
        {
          count -= 128;
          if(count == 0 || x + count > w)
            fail("HDR: corrupt run");
          const uint8_t v = uint8_t(r.get());
          for(int i = 0; i < count; ++i)
            scan[size_t(x++) * 4 + k] = v;
        }
        else
        {
          if(count == 0 || x + count > w)
            fail("HDR: corrupt run");
          for(int i = 0; i < count; ++i)
            scan[size_t(x++) * 4 + k] = uint8_t(r.get());
        }
      }
    }
    for(int x = 0; x < w; ++x)
      rgbe_to_float(&scan[size_t(x) * 4], &rgba[(size_t(y) * w + x) * 4]);
  }
}
}  // namespace

extern "C" {

int pt_hdr_load(const char* path, float** out_rgba32f, int* out_width, int* out_height, char* err, size_t err_len)
{
  auto report = [&](const char* m) {
    if(err && err_len)
      std::snprintf(err, err_len, "%s", m);
    return PT_ERR_INVALID;
  };
  if(!path || !out_rgba32f || !out_width || !out_height)
    return report("pt_hdr_load: null argument");
  *out_rgba32f = nullptr;
  FILE* f      = std::fopen(path, "rb");
  if(!f)
    return report("pt_hdr_load: cannot open the file");
  std::vector<uint8_t> file;
  uint8_t              buf[65536];
  size_t               n;
  while((n = std::fread(buf, 1, sizeof(buf), f)) > 0)
  {
    file.insert(file.end(), buf, buf + n);
    if(file.size() > (size_t(1) << 32))
      break;
  }
  std::fclose(f);
  try
  {
    std::vector<float> rgba;
    int                w = 0, h = 0;
    decode(file, rgba, w, h);
    float* out = static_cast<float*>(std::malloc(rgba.size() * sizeof(float)));  // malloc: the reference frees with stbi_image_free == free
    if(!out)
      return report("pt_hdr_load: out of memory");
    std::memcpy(out, rgba.data(), rgba.size() * sizeof(float));
    *out_rgba32f = out;
    *out_width   = w;
    *out_height  = h;
    return PT_OK;
  }
  catch(const Fail& e)
  {
    return report(e.msg.c_str());
  }
  catch(const std::exception& e)
  {
    return report(e.what());
  }
}

void pt_hdr_free(float* rgba32f) { std::free(rgba32f); }
}
