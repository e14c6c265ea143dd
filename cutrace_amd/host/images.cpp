// images.cpp — the three image writers either side of the hot path's outputs.
//
// Quantisation follows the reference exactly (inc/images.hpp:26-88):
//   depth  : finite ? (byte)(255*(max-v)/max) : 0, replicated to 3 channels  (:27-29)
//   normal : norm<=1e-6 ? 0 : (byte)(255*(0.5+0.5*normalized))                (:48-54)
//   colour : clamp to [0,1] then (byte)(255*c) — truncation                   (:73-76)
// The JPEG container is a from-scratch baseline (sequential DCT, Huffman, 4:4:4)
// encoder using the standard Annex-K tables scaled IJG-style for `quality`; the
// reference delegates to stb_image_write at quality 90 (:39,64,86).  The JPEG
// BYTES are not expected to match stb's (parity is judged on the float buffers
// and on the quantised planes); the decoded image is the same picture.
//
// Compile with -ffp-contract=off (quantisation is float arithmetic).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../include/cutrace_host.h"

namespace {

typedef unsigned char byte;

inline float norm3(float x, float y, float z) { return sqrtf(x * x + y * y + z * z); }

// ---------------- baseline JPEG encoder ----------------
const byte ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                         41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                         30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
const byte Q_LUMA[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40,  57,
                         69, 56, 14, 17, 22,  29,  51,  87,  80, 62, 18, 22, 37,  56,  68,  109, 103, 77, 24, 35, 55,  64,
                         81, 104, 113, 92, 49, 64,  78,  87,  103, 121, 120, 101, 72, 92,  95,  98,  112, 100, 103, 99};
const byte Q_CHROMA[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                           99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                           99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
// Annex K.3 Huffman specifications
const byte DC_L_BITS[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const byte DC_L_VAL[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const byte DC_C_BITS[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const byte DC_C_VAL[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const byte AC_L_BITS[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const byte AC_L_VAL[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const byte AC_C_BITS[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const byte AC_C_VAL[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct huff {
  unsigned short code[256];
  byte len[256];
  void build(const byte *bits, const byte *vals) {
    memset(len, 0, sizeof(len));
    unsigned c = 0;
    int k = 0;
    for (int l = 1; l <= 16; l++) {
      for (int i = 0; i < bits[l - 1]; i++) {
        code[vals[k]] = (unsigned short)c++;
        len[vals[k]] = (byte)l;
        k++;
      }
      c <<= 1;
    }
  }
};

struct bitwriter {
  std::vector<byte> &out;
  unsigned acc = 0;
  int n = 0;
  explicit bitwriter(std::vector<byte> &o) : out(o) {}
  void put(unsigned code, int len) {
    acc = (acc << len) | (code & ((1u << len) - 1));
    n += len;
    while (n >= 8) {
      byte b = (byte)((acc >> (n - 8)) & 0xFF);
      out.push_back(b);
      if (b == 0xFF) out.push_back(0);
      n -= 8;
    }
  }
  void flush() {
    if (n > 0) put(0x7F, 8 - n);  // pad with 1-bits
  }
};

// the DCT basis, built once (a function-local static object: its construction is thread-safe — the three images of a frame
// are written by three threads, and each writer transforms its blocks on several more)
struct DctBasis {
  float C[8][8];
  DctBasis() {
    for (int u = 0; u < 8; u++)
      for (int x = 0; x < 8; x++) C[u][x] = (float)((u == 0 ? sqrt(0.125) : 0.5) * cos((2 * x + 1) * u * M_PI / 16.0));
  }
};

void fdct8x8(const float *in, float *out) {
  // separable direct DCT-II (clarity over speed; the speed comes from doing the blocks in parallel)
  static const DctBasis B;
  const float (*C)[8] = B.C;
  float tmp[64];
  for (int y = 0; y < 8; y++)
    for (int u = 0; u < 8; u++) {
      float s = 0;
      for (int x = 0; x < 8; x++) s += C[u][x] * in[y * 8 + x];
      tmp[y * 8 + u] = s;
    }
  for (int u = 0; u < 8; u++)
    for (int v = 0; v < 8; v++) {
      float s = 0;
      for (int y = 0; y < 8; y++) s += C[v][y] * tmp[y * 8 + u];
      out[v * 8 + u] = s;
    }
}

void put_marker(std::vector<byte> &o, byte m) { o.push_back(0xFF); o.push_back(m); }
void put16(std::vector<byte> &o, unsigned v) { o.push_back((byte)(v >> 8)); o.push_back((byte)(v & 0xFF)); }

// A block in two steps, so that the first — transform and quantisation, independent from block to block and most of the
// work — can run on several threads, while the second — entropy coding, where every block's DC value is predicted from the
// previous block's and the bits are packed one after the other — runs in order afterwards.  Same arithmetic as one step.
void quantise_block(const float *blk, const byte *q, short *z) {
  float f[64];
  fdct8x8(blk, f);
  for (int i = 0; i < 64; i++) {
    float v = f[ZIGZAG[i]] / (float)q[ZIGZAG[i]];
    z[i] = (short)(int)(v < 0 ? v - 0.5f : v + 0.5f);
  }
}

void entropy_block(bitwriter &bw, const short *z, int &dc_prev, const huff &dc, const huff &ac) {
  auto magnitude = [](int v, int &bits) {
    int a = v < 0 ? -v : v, n = 0;
    while (a) { n++; a >>= 1; }
    bits = v < 0 ? v - 1 + (1 << n) : v;  // one's-complement style negative
    return n;
  };
  int diff = z[0] - dc_prev, bits;
  dc_prev = z[0];
  int n = magnitude(diff, bits);
  bw.put(dc.code[n], dc.len[n]);
  if (n) bw.put((unsigned)bits, n);
  int last = 63;
  while (last > 0 && z[last] == 0) last--;
  int run = 0;
  for (int i = 1; i <= last; i++) {
    if (z[i] == 0) { run++; continue; }
    while (run >= 16) { bw.put(ac.code[0xF0], ac.len[0xF0]); run -= 16; }
    n = magnitude(z[i], bits);
    int sym = (run << 4) | n;
    bw.put(ac.code[sym], ac.len[sym]);
    bw.put((unsigned)bits, n);
    run = 0;
  }
  if (last != 63) bw.put(ac.code[0], ac.len[0]);  // EOB
}

}  // namespace

extern "C" {

void ctr_quantise_depth(const float *depth, uint64_t n, float max_d, unsigned char *out) {
  for (uint64_t i = 0; i < n; i++) {
    float v = depth[i];
    byte b = std::isfinite(v) ? (byte)(255 * (max_d - v) / max_d) : 0;
    out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = b;
  }
}

void ctr_quantise_normal(const float *n3, uint64_t n, unsigned char *out) {
  for (uint64_t i = 0; i < n; i++) {
    float x = n3[3 * i], y = n3[3 * i + 1], z = n3[3 * i + 2];
    float len = norm3(x, y, z);
    if (len <= 1e-6) { out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = 0; continue; }
    float f = 1.0f / len;  // normalized(): v * (1/norm), vector.hpp:77-79
    float nx = 0.5f + 0.5f * (f * x), ny = 0.5f + 0.5f * (f * y), nz = 0.5f + 0.5f * (f * z);
    out[3 * i] = (byte)(255 * nx);
    out[3 * i + 1] = (byte)(255 * ny);
    out[3 * i + 2] = (byte)(255 * nz);
  }
}

void ctr_quantise_color(const float *c3, uint64_t n, unsigned char *out) {
  for (uint64_t i = 0; i < 3 * n; i++) {
    float v = c3[i];
    float lo = (0.0f < v) ? v : 0.0f;    // std::max(0.0f, v)
    float c = (lo < 1.0f) ? lo : 1.0f;   // std::min(1.0f, lo)
    out[i] = (byte)(255 * c);
  }
}

int ctr_write_jpg(const char *path, int w, int h, const unsigned char *rgb, int quality) {
  if (!path || !rgb || w <= 0 || h <= 0 || w > 65535 || h > 65535) return CTR_E_INVALID;
  if (quality < 1) quality = 1;
  if (quality > 100) quality = 100;
  int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
  byte ql[64], qc[64];
  for (int i = 0; i < 64; i++) {
    int a = (Q_LUMA[i] * scale + 50) / 100, b = (Q_CHROMA[i] * scale + 50) / 100;
    ql[i] = (byte)(a < 1 ? 1 : a > 255 ? 255 : a);
    qc[i] = (byte)(b < 1 ? 1 : b > 255 ? 255 : b);
  }
  std::vector<byte> o;
  o.reserve((size_t)w * h / 2 + 1024);
  put_marker(o, 0xD8);
  put_marker(o, 0xE0); put16(o, 16);
  const byte jfif[] = {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
  o.insert(o.end(), jfif, jfif + 14);
  put_marker(o, 0xDB); put16(o, 2 + 65 * 2);
  o.push_back(0); for (int i = 0; i < 64; i++) o.push_back(ql[ZIGZAG[i]]);
  o.push_back(1); for (int i = 0; i < 64; i++) o.push_back(qc[ZIGZAG[i]]);
  put_marker(o, 0xC0); put16(o, 17); o.push_back(8); put16(o, (unsigned)h); put16(o, (unsigned)w); o.push_back(3);
  o.push_back(1); o.push_back(0x11); o.push_back(0);
  o.push_back(2); o.push_back(0x11); o.push_back(1);
  o.push_back(3); o.push_back(0x11); o.push_back(1);
  put_marker(o, 0xC4); put16(o, 2 + (1 + 16 + 12) * 2 + (1 + 16 + 162) * 2);
  auto dht = [&](byte id, const byte *bits, const byte *vals, int nv) {
    o.push_back(id);
    o.insert(o.end(), bits, bits + 16);
    o.insert(o.end(), vals, vals + nv);
  };
  dht(0x00, DC_L_BITS, DC_L_VAL, 12);
  dht(0x10, AC_L_BITS, AC_L_VAL, 162);
  dht(0x01, DC_C_BITS, DC_C_VAL, 12);
  dht(0x11, AC_C_BITS, AC_C_VAL, 162);
  put_marker(o, 0xDA); put16(o, 12); o.push_back(3);
  o.push_back(1); o.push_back(0x00);
  o.push_back(2); o.push_back(0x11);
  o.push_back(3); o.push_back(0x11);
  o.push_back(0); o.push_back(63); o.push_back(0);

  huff hdl, hal, hdc, hac;
  hdl.build(DC_L_BITS, DC_L_VAL); hal.build(AC_L_BITS, AC_L_VAL);
  hdc.build(DC_C_BITS, DC_C_VAL); hac.build(AC_C_BITS, AC_C_VAL);
  // step 1, in parallel over the rows of blocks: colour transform, DCT, quantisation -> 3 x 64 coefficients per block
  const int nbx = (w + 7) / 8, nby = (h + 7) / 8;
  std::vector<short> coef((size_t)nbx * nby * 3 * 64);
  auto rows = [&](int first, int step) {
    float Y[64], Cb[64], Cr[64];
    for (int br = first; br < nby; br += step)
      for (int bc = 0; bc < nbx; bc++) {
        const int by = br * 8, bx = bc * 8;
        for (int yy = 0; yy < 8; yy++)
          for (int xx = 0; xx < 8; xx++) {
            int y = by + yy < h ? by + yy : h - 1, x = bx + xx < w ? bx + xx : w - 1;
            const byte *p = rgb + 3 * ((size_t)y * w + x);
            float r = p[0], g = p[1], b = p[2];
            Y[yy * 8 + xx] = 0.299f * r + 0.587f * g + 0.114f * b - 128.0f;
            Cb[yy * 8 + xx] = -0.168736f * r - 0.331264f * g + 0.5f * b;
            Cr[yy * 8 + xx] = 0.5f * r - 0.418688f * g - 0.081312f * b;
          }
        short *z = &coef[((size_t)br * nbx + bc) * 3 * 64];
        quantise_block(Y, ql, z);
        quantise_block(Cb, qc, z + 64);
        quantise_block(Cr, qc, z + 128);
      }
  };
  {
    unsigned hw = std::thread::hardware_concurrency();
    if (const char *e = getenv("CUTRACE_JPEG_THREADS")) hw = (unsigned)std::max(1, atoi(e));
    const int T = std::max(1, std::min({(int)(hw ? hw : 1u), 16, nby / 4 + 1}));
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(rows, t, T);
    rows(0, T);
    for (std::thread &x : th) x.join();
  }
  // step 2, in order: entropy coding
  bitwriter bw(o);
  int dcy = 0, dcb = 0, dcr = 0;
  for (size_t k = 0; k < (size_t)nbx * nby; k++) {
    const short *z = &coef[k * 3 * 64];
    entropy_block(bw, z, dcy, hdl, hal);
    entropy_block(bw, z + 64, dcb, hdc, hac);
    entropy_block(bw, z + 128, dcr, hdc, hac);
  }
  bw.flush();
  put_marker(o, 0xD9);
  FILE *f = fopen(path, "wb");
  if (!f) return CTR_E_IO;
  size_t wr = fwrite(o.data(), 1, o.size(), f);
  fclose(f);
  return wr == o.size() ? CTR_OK : CTR_E_IO;
}

int ctr_write_depth_map(const char *path, const float *depth, uint64_t w, uint64_t h, float max_d) {
  std::vector<byte> px(3 * w * h);
  ctr_quantise_depth(depth, w * h, max_d, px.data());
  return ctr_write_jpg(path, (int)w, (int)h, px.data(), 90);
}
int ctr_write_normal_map(const char *path, const float *normal3, uint64_t w, uint64_t h) {
  std::vector<byte> px(3 * w * h);
  ctr_quantise_normal(normal3, w * h, px.data());
  return ctr_write_jpg(path, (int)w, (int)h, px.data(), 90);
}
int ctr_write_colorized(const char *path, const float *color3, uint64_t w, uint64_t h) {
  std::vector<byte> px(3 * w * h);
  ctr_quantise_color(color3, w * h, px.data());
  return ctr_write_jpg(path, (int)w, (int)h, px.data(), 90);
}

}  // extern "C"
