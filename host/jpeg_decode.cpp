// jpeg_decode.cpp — baseline and progressive Huffman JPEG decoder (ITU-T T.81) for the skybox faces.
//
// Replaces the reference's `stbi_load(path, &w, &h, &c, STBI_rgb_alpha)` (src/main.cpp:2073-2080,
// vendored stb_image.h v2.27) on the host side of the cube-map upload.  Independent implementation
// of the standard; the three places where T.81 leaves numerics open follow the same published
// recipes stb_image uses so that the decoded texels — and therefore the rendered sky — are the
// reference's, which tests/test_host.py pins byte-for-byte with tests/golden/ingest_golden.json:
//   * inverse DCT: Loeffler-Ligtenberg-Moschytz integer IDCT (IJG "islow"), 12-bit constants,
//     2 extra bits after the column pass, +128 level shift folded into the row pass rounding;
//   * chroma up-sampling: triangle filter (3:1 weights; 9:3:3:1 for 2x2) with centred samples;
//   * YCbCr -> RGB: JFIF matrix in 20-bit fixed point built from 12-bit rounded constants.
// Supported: SOF0/SOF1/SOF2, 8-bit, 1 or 3 components, sampling factors 1..2 (hv 1x1, 2x1, 1x2,
// 2x2), restart intervals.  Output is always RGBA8 with alpha 255 (STBI_rgb_alpha).
#include "jpeg_decode.h"

#include <cstdio>
#include <cstring>

namespace rtjpeg {
namespace {

const uint8_t kZigZag[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                  6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                  39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                  // guard entries for corrupt run lengths
                                  63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huff {
  bool present = false;
  uint8_t vals[256];
  int mincode[18], maxcode[18], valptr[18];
  uint8_t fast_len[512];
  uint8_t fast_val[512];
  void build(const uint8_t counts[16], const uint8_t* symbols, int n) {
    memcpy(vals, symbols, (size_t)n);
    memset(fast_len, 0, sizeof(fast_len));
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
      valptr[l] = k;
      mincode[l] = code;
      for (int i = 0; i < counts[l - 1]; i++, k++, code++) {
        if (l <= 9) {
          int first = code << (9 - l), cnt = 1 << (9 - l);
          for (int j = 0; j < cnt; j++) { fast_len[first + j] = (uint8_t)l; fast_val[first + j] = vals[k]; }
        }
      }
      maxcode[l] = counts[l - 1] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7FFFFFFF;
    present = true;
  }
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0;
  int td = 0, ta = 0;          // Huffman table selectors of the current scan
  int bw = 0, bh = 0;          // allocated blocks (whole MCUs)
  int cw = 0, ch = 0;          // blocks actually covered by a non-interleaved scan
  int pw = 0, ph = 0;          // plane size in samples (bw*8, bh*8)
  int sw = 0, sh = 0;          // real sample extent: ceil(W*h/hmax), ceil(H*v/vmax)
  int dc_pred = 0;
  std::vector<int16_t> coef;   // bw*bh*64, natural order
  std::vector<uint8_t> plane;
};

struct Decoder {
  const uint8_t* p; const uint8_t* end;
  std::string* err;
  int W = 0, H = 0, ncomp = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
  bool progressive = false;
  uint16_t qt[4][64];          // natural order
  bool qt_present[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  Component comp[3];
  int restart_interval = 0;
  // entropy-coded segment reader
  uint64_t bitbuf = 0; int bitcnt = 0; int marker = 0;
  int eobrun = 0;

  bool fail(const char* m) { if (err->empty()) *err = m; return false; }

  // ---- bit reader with byte stuffing -------------------------------------------------------
  void fill() {
    while (bitcnt <= 56) {
      int b = 0;
      if (!marker && p < end) {
        b = *p++;
        if (b == 0xFF) {
          int c = p < end ? *p : 0;
          while (c == 0xFF && p + 1 < end) { p++; c = *p; }  // fill bytes
          if (c == 0) p++;                                    // stuffed zero: data byte 0xFF
          else { marker = c; p++; b = 0; }                    // marker: feed zeros from here on
        }
      }
      bitbuf |= (uint64_t)b << (56 - bitcnt);
      bitcnt += 8;
    }
  }
  inline int peek(int n) { if (bitcnt < n) fill(); return (int)(bitbuf >> (64 - n)); }
  inline void skip(int n) { bitbuf <<= n; bitcnt -= n; }
  inline int getbits(int n) { if (n == 0) return 0; int v = peek(n); skip(n); return v; }
  inline int getbit() { return getbits(1); }
  inline int receive_extend(int n) {
    if (n == 0) return 0;
    int v = getbits(n);
    return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
  }
  inline int decode_huff(const Huff& h) {
    int look = peek(16);
    int l = h.fast_len[look >> 7];
    if (l) { skip(l); return h.fast_val[look >> 7]; }
    for (l = 10; l <= 16; l++) {
      int code = look >> (16 - l);
      if (code <= h.maxcode[l]) { skip(l); return h.vals[(h.valptr[l] + code - h.mincode[l]) & 255]; }
    }
    skip(16);
    return 0;
  }
  void reset_entropy() { bitbuf = 0; bitcnt = 0; marker = 0; eobrun = 0; for (int c = 0; c < ncomp; c++) comp[c].dc_pred = 0; }

  // ---- marker segments ------------------------------------------------------------------------
  static int be16(const uint8_t* q) { return (q[0] << 8) | q[1]; }

  bool parse_dqt(const uint8_t* s, int len) {
    while (len > 0) {
      int pq = s[0] >> 4, tq = s[0] & 15;
      if (tq > 3) return fail("bad DQT table id");
      s++; len--;
      for (int i = 0; i < 64; i++) {
        int q = pq ? be16(s + 2 * i) : s[i];
        qt[tq][kZigZag[i]] = (uint16_t)q;
      }
      s += pq ? 128 : 64; len -= pq ? 128 : 64;
      qt_present[tq] = true;
    }
    return len == 0 ? true : fail("bad DQT length");
  }
  bool parse_dht(const uint8_t* s, int len) {
    while (len > 0) {
      if (len < 17) return fail("bad DHT length");
      int tc = s[0] >> 4, th = s[0] & 15;
      if (tc > 1 || th > 3) return fail("bad DHT header");
      int n = 0;
      for (int i = 0; i < 16; i++) n += s[1 + i];
      if (n > 256 || len < 17 + n) return fail("bad DHT counts");
      (tc ? ac[th] : dc[th]).build(s + 1, s + 17, n);
      s += 17 + n; len -= 17 + n;
    }
    return true;
  }
  bool parse_sof(const uint8_t* s, int len, int kind) {
    if (len < 6) return fail("bad SOF length");
    if (s[0] != 8) return fail("only 8-bit JPEG is supported");
    H = be16(s + 1); W = be16(s + 3); ncomp = s[5];
    if (W <= 0 || H <= 0) return fail("bad image size");
    if (ncomp != 1 && ncomp != 3) return fail("only 1- or 3-component JPEG is supported");
    if (len < 6 + 3 * ncomp) return fail("bad SOF length");
    progressive = kind == 2;
    hmax = vmax = 1;
    for (int c = 0; c < ncomp; c++) {
      comp[c].id = s[6 + 3 * c]; comp[c].h = s[7 + 3 * c] >> 4; comp[c].v = s[7 + 3 * c] & 15; comp[c].tq = s[8 + 3 * c];
      if (comp[c].h < 1 || comp[c].h > 2 || comp[c].v < 1 || comp[c].v > 2 || comp[c].tq > 3) return fail("unsupported sampling factors");
      if (comp[c].h > hmax) hmax = comp[c].h;
      if (comp[c].v > vmax) vmax = comp[c].v;
    }
    mcux = (W + 8 * hmax - 1) / (8 * hmax); mcuy = (H + 8 * vmax - 1) / (8 * vmax);
    for (int c = 0; c < ncomp; c++) {
      Component& k = comp[c];
      k.bw = mcux * k.h; k.bh = mcuy * k.v;
      k.sw = (W * k.h + hmax - 1) / hmax; k.sh = (H * k.v + vmax - 1) / vmax;
      k.cw = (k.sw + 7) / 8; k.ch = (k.sh + 7) / 8;
      k.pw = k.bw * 8; k.ph = k.bh * 8;
      k.coef.assign((size_t)k.bw * k.bh * 64, 0);
    }
    return true;
  }

  // ---- block decoders ---------------------------------------------------------------------------
  bool block_baseline(Component& k, int16_t* b) {
    const Huff& hd = dc[k.td]; const Huff& ha = ac[k.ta];
    int t = decode_huff(hd);
    int diff = receive_extend(t & 15);
    k.dc_pred += diff;
    b[0] = (int16_t)k.dc_pred;
    for (int i = 1; i < 64;) {
      int rs = decode_huff(ha), r = rs >> 4, s = rs & 15;
      if (s == 0) { if (r != 15) break; i += 16; continue; }
      i += r;
      if (i > 63) return fail("bad AC run");
      b[kZigZag[i]] = (int16_t)receive_extend(s);
      i++;
    }
    return true;
  }
  bool block_dc_prog(Component& k, int16_t* b, int ah, int al) {
    if (ah == 0) {
      int t = decode_huff(dc[k.td]);
      int diff = receive_extend(t & 15);
      k.dc_pred += diff;
      b[0] = (int16_t)(k.dc_pred * (1 << al));
    } else if (getbit()) b[0] = (int16_t)(b[0] + (1 << al));
    return true;
  }
  bool block_ac_prog(Component& k, int16_t* b, int ss, int se, int ah, int al) {
    const Huff& ha = ac[k.ta];
    if (ah == 0) {
      if (eobrun) { eobrun--; return true; }
      for (int i = ss; i <= se;) {
        int rs = decode_huff(ha), r = rs >> 4, s = rs & 15;
        if (s == 0) {
          if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += getbits(r); break; }
          i += 16;
        } else {
          i += r;
          if (i > 63) return fail("bad AC run");
          b[kZigZag[i]] = (int16_t)(receive_extend(s) * (1 << al));
          i++;
        }
      }
      return true;
    }
    // successive-approximation refinement
    const int p1 = 1 << al, m1 = -(1 << al);
    int i = ss;
    auto refine = [&](int16_t& c) {
      if (getbit() && (c & p1) == 0) c = (int16_t)(c + (c >= 0 ? p1 : m1));
    };
    if (eobrun == 0) {
      for (; i <= se; i++) {
        int rs = decode_huff(ha), r = rs >> 4, s = rs & 15;
        int val = 0;
        if (s == 0) {
          if (r < 15) { eobrun = (1 << r); if (r) eobrun += getbits(r); break; }
        } else {
          if (s != 1) return fail("bad refinement symbol");
          val = getbit() ? p1 : m1;
        }
        for (; i <= se; i++) {
          int16_t& c = b[kZigZag[i]];
          if (c != 0) refine(c);
          else { if (r == 0) { if (val) c = (int16_t)val; break; } r--; }
        }
      }
    }
    if (eobrun > 0) {
      for (; i <= se; i++) { int16_t& c = b[kZigZag[i]]; if (c != 0) refine(c); }
      eobrun--;
    }
    return true;
  }

  bool handle_restart(int& todo) {
    if (restart_interval == 0) return true;
    if (--todo > 0) return true;
    // expect RSTn
    if (!marker) { bitcnt = 0; bitbuf = 0; fill(); }
    if (marker >= 0xD0 && marker <= 0xD7) { reset_entropy(); todo = restart_interval; return true; }
    if (marker == 0) { // marker may still be ahead in the stream
      while (p + 1 < end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) { if (p[0] == 0xFF && p[1] != 0 && p[1] != 0xFF) break; p++; }
      if (p + 1 < end && p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7) { p += 2; reset_entropy(); todo = restart_interval; return true; }
    }
    todo = restart_interval;  // no restart marker: let the caller hit the end of scan
    return true;
  }

  bool decode_scan(const uint8_t* s, int len) {
    int ns = s[0];
    if (ns < 1 || ns > ncomp || len < 4 + 2 * ns) return fail("bad SOS header");
    int order[3];
    for (int i = 0; i < ns; i++) {
      int id = s[1 + 2 * i], which = -1;
      for (int c = 0; c < ncomp; c++) if (comp[c].id == id) which = c;
      if (which < 0) return fail("SOS references an unknown component");
      comp[which].td = s[2 + 2 * i] >> 4; comp[which].ta = s[2 + 2 * i] & 15;
      if (comp[which].td > 3 || comp[which].ta > 3) return fail("bad table selector");
      order[i] = which;
    }
    int ss = s[1 + 2 * ns], se = s[2 + 2 * ns], ah = s[3 + 2 * ns] >> 4, al = s[3 + 2 * ns] & 15;
    if (!progressive) { ss = 0; se = 63; ah = al = 0; }
    else if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13 || (ss == 0 && se != 0) || (ss > 0 && ns != 1)) return fail("bad progressive scan parameters");
    reset_entropy();
    int todo = restart_interval ? restart_interval : 0x7FFFFFFF;
    auto one = [&](Component& k, int bx, int by) -> bool {
      int16_t* b = &k.coef[((size_t)by * k.bw + bx) * 64];
      if (!progressive) return block_baseline(k, b);
      if (ss == 0) return block_dc_prog(k, b, ah, al);
      return block_ac_prog(k, b, ss, se, ah, al);
    };
    if (ns == 1) {
      Component& k = comp[order[0]];
      for (int by = 0; by < k.ch; by++)
        for (int bx = 0; bx < k.cw; bx++) {
          if (!one(k, bx, by)) return false;
          if (!handle_restart(todo)) return false;
        }
    } else {
      for (int my = 0; my < mcuy; my++)
        for (int mx = 0; mx < mcux; mx++) {
          for (int i = 0; i < ns; i++) {
            Component& k = comp[order[i]];
            for (int y = 0; y < k.v; y++)
              for (int x = 0; x < k.h; x++)
                if (!one(k, mx * k.h + x, my * k.v + y)) return false;
          }
          if (!handle_restart(todo)) return false;
        }
    }
    // position p at the marker that ended the entropy-coded segment
    if (marker) { p -= 2; }
    else {
      while (p + 1 < end && !(p[0] == 0xFF && p[1] != 0 && p[1] != 0xFF && !(p[1] >= 0xD0 && p[1] <= 0xD7))) p++;
    }
    return true;
  }

  // ---- reconstruction ---------------------------------------------------------------------------
  static inline uint8_t clamp8(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

  // LL&M integer IDCT: CONST_BITS = 12, PASS1_BITS = 2
  static void idct8x8(const int16_t* in, uint8_t* out, int stride) {
    // 12-bit constants.  The five that enter with a minus sign are int(-c*4096 + 0.5) truncated toward
    // zero (1597, 3685, 7567, 8034, 10497 rather than the nearest 1598, 3686, 7568, 8035, 10498): that
    // is the table the reference's decoder executes, and matching it is what makes the faces identical.
    const int C_0_298 = 1223, C_0_390 = 1597, C_0_541 = 2217, C_0_765 = 3135, C_0_899 = 3685, C_1_175 = 4816, C_1_501 = 6149,
              C_1_847 = 7567, C_1_961 = 8034, C_2_053 = 8410, C_2_562 = 10497, C_3_072 = 12586;
    int ws[64];
    auto pass = [&](int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7, int* even, int* odd) {
      int z1 = (s2 + s6) * C_0_541;
      int e2 = z1 - s6 * C_1_847, e3 = z1 + s2 * C_0_765;
      int e0 = (s0 + s4) * 4096, e1 = (s0 - s4) * 4096;
      even[0] = e0 + e3; even[3] = e0 - e3; even[1] = e1 + e2; even[2] = e1 - e2;
      int a = s7, b = s5, c = s3, d = s1;
      int z3 = a + c, z4 = b + d, z1b = a + d, z2b = b + c;
      int z5 = (z3 + z4) * C_1_175;
      a *= C_0_298; b *= C_2_053; c *= C_3_072; d *= C_1_501;
      int y1 = z5 - z1b * C_0_899, y2 = z5 - z2b * C_2_562;
      z3 = -z3 * C_1_961; z4 = -z4 * C_0_390;
      odd[3] = d + y1 + z4; odd[2] = c + y2 + z3; odd[1] = b + y2 + z4; odd[0] = a + y1 + z3;
    };
    for (int c = 0; c < 8; c++) {
      int ev[4], od[4];
      pass(in[c], in[8 + c], in[16 + c], in[24 + c], in[32 + c], in[40 + c], in[48 + c], in[56 + c], ev, od);
      for (int k = 0; k < 4; k++) {
        ws[8 * k + c] = (ev[k] + 512 + od[3 - k]) >> 10;
        ws[8 * (7 - k) + c] = (ev[k] + 512 - od[3 - k]) >> 10;
      }
    }
    for (int r = 0; r < 8; r++) {
      const int* w = ws + 8 * r;
      int ev[4], od[4];
      pass(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], ev, od);
      uint8_t* o = out + (size_t)r * stride;
      const int bias = 65536 + (128 << 17);
      for (int k = 0; k < 4; k++) {
        o[k] = clamp8((ev[k] + bias + od[3 - k]) >> 17);
        o[7 - k] = clamp8((ev[k] + bias - od[3 - k]) >> 17);
      }
    }
  }

  bool reconstruct(Image& img) {
    for (int c = 0; c < ncomp; c++) {
      Component& k = comp[c];
      if (!qt_present[k.tq]) return fail("missing quantisation table");
      k.plane.assign((size_t)k.pw * k.ph, 0);
      int16_t tmp[64];
      for (int by = 0; by < k.bh; by++)
        for (int bx = 0; bx < k.bw; bx++) {
          const int16_t* b = &k.coef[((size_t)by * k.bw + bx) * 64];
          for (int i = 0; i < 64; i++) tmp[i] = (int16_t)(b[i] * qt[k.tq][i]);
          idct8x8(tmp, &k.plane[(size_t)by * 8 * k.pw + (size_t)bx * 8], k.pw);
        }
    }
    img.w = W; img.h = H;
    img.rgba.assign((size_t)W * H * 4, 255);
    std::vector<uint8_t> line[3];
    for (int c = 0; c < ncomp; c++) line[c].resize((size_t)W + 8);
    for (int y = 0; y < H; y++) {
      const uint8_t* row[3] = {nullptr, nullptr, nullptr};
      for (int c = 0; c < ncomp; c++) {
        Component& k = comp[c];
        int hs = hmax / k.h, vs = vmax / k.v;
        const uint8_t *nr, *fr;
        if (vs == 1) { nr = fr = &k.plane[(size_t)y * k.pw]; }
        else {
          int n = y >> 1, f = (y & 1) ? n + 1 : n - 1;
          if (f < 0) f = 0;
          if (f > k.sh - 1) f = k.sh - 1;
          if (n > k.sh - 1) n = k.sh - 1;
          nr = &k.plane[(size_t)n * k.pw]; fr = &k.plane[(size_t)f * k.pw];
        }
        uint8_t* o = line[c].data();
        const int wl = (W + hs - 1) / hs;
        if (hs == 1 && vs == 1) { row[c] = nr; continue; }
        if (hs == 1) { for (int i = 0; i < wl; i++) o[i] = (uint8_t)((3 * nr[i] + fr[i] + 2) >> 2); }
        else if (vs == 1) {
          if (wl == 1) { o[0] = o[1] = nr[0]; }
          else {
            o[0] = nr[0]; o[1] = (uint8_t)((nr[0] * 3 + nr[1] + 2) >> 2);
            int i;
            for (i = 1; i < wl - 1; i++) { int n3 = 3 * nr[i] + 2; o[2 * i] = (uint8_t)((n3 + nr[i - 1]) >> 2); o[2 * i + 1] = (uint8_t)((n3 + nr[i + 1]) >> 2); }
            o[2 * i] = (uint8_t)((nr[wl - 2] * 3 + nr[wl - 1] + 2) >> 2); o[2 * i + 1] = nr[wl - 1];
          }
        } else {
          if (wl == 1) { o[0] = o[1] = (uint8_t)((3 * nr[0] + fr[0] + 2) >> 2); }
          else {
            int t1 = 3 * nr[0] + fr[0], t0;
            o[0] = (uint8_t)((t1 + 2) >> 2);
            for (int i = 1; i < wl; i++) {
              t0 = t1; t1 = 3 * nr[i] + fr[i];
              o[2 * i - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
              o[2 * i] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
            }
            o[2 * wl - 1] = (uint8_t)((t1 + 2) >> 2);
          }
        }
        row[c] = o;
      }
      uint8_t* out = &img.rgba[(size_t)y * W * 4];
      if (ncomp == 1) {
        for (int x = 0; x < W; x++) { out[4 * x] = out[4 * x + 1] = out[4 * x + 2] = row[0][x]; }
      } else {
        // JFIF YCbCr -> RGB, 20-bit fixed point from 12-bit rounded constants
        const int CR_R = 5743 << 8, CR_G = -(2925 << 8), CB_G = -(1410 << 8), CB_B = 7258 << 8;
        for (int x = 0; x < W; x++) {
          int yf = (row[0][x] << 20) + (1 << 19);
          int cb = row[1][x] - 128, cr = row[2][x] - 128;
          int r = yf + cr * CR_R;
          int g = yf + cr * CR_G + (int)((unsigned)(cb * CB_G) & 0xffff0000u);
          int b = yf + cb * CB_B;
          out[4 * x] = clamp8(r >> 20); out[4 * x + 1] = clamp8(g >> 20); out[4 * x + 2] = clamp8(b >> 20);
        }
      }
    }
    return true;
  }

  bool run(Image& img) {
    if (end - p < 4 || p[0] != 0xFF || p[1] != 0xD8) return fail("not a JPEG (no SOI)");
    p += 2;
    bool have_sof = false, have_scan = false;
    while (p + 4 <= end) {
      if (p[0] != 0xFF) { p++; continue; }
      int m = p[1];
      if (m == 0xFF) { p++; continue; }
      if (m == 0xD9) break;
      if (m == 0x01 || (m >= 0xD0 && m <= 0xD7) || m == 0x00) { p += 2; continue; }
      int len = be16(p + 2);
      if (len < 2 || p + 2 + len > end) return fail("truncated marker segment");
      const uint8_t* s = p + 4;
      p += 2 + len;
      if (m == 0xDB) { if (!parse_dqt(s, len - 2)) return false; }
      else if (m == 0xC4) { if (!parse_dht(s, len - 2)) return false; }
      else if (m == 0xC0 || m == 0xC1 || m == 0xC2) { if (have_sof) return fail("multiple SOF"); if (!parse_sof(s, len - 2, m - 0xC0)) return false; have_sof = true; }
      else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) return fail("unsupported JPEG process (lossless/arithmetic/hierarchical)");
      else if (m == 0xDD) { if (len < 4) return fail("bad DRI"); restart_interval = be16(s); }
      else if (m == 0xDA) {
        if (!have_sof) return fail("SOS before SOF");
        if (!decode_scan(s, len - 2)) return false;
        have_scan = true;
      }
    }
    if (!have_sof || !have_scan) return fail("no image data");
    return reconstruct(img);
  }
};

}  // namespace

bool decode_memory(const uint8_t* data, size_t n, Image& out, std::string& err) {
  err.clear();
  Decoder d;
  d.p = data; d.end = data + n; d.err = &err;
  return d.run(out);
}

bool decode_file(const char* path, Image& out, std::string& err) {
  FILE* f = fopen(path, "rb");
  if (!f) { err = std::string("cannot open ") + path; return false; }
  std::vector<uint8_t> buf;
  uint8_t tmp[65536];
  size_t n;
  while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
  fclose(f);
  return decode_memory(buf.data(), buf.size(), out, err);
}

}  // namespace rtjpeg
