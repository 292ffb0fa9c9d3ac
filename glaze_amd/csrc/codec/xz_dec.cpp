// .xz container + LZMA2 + LZMA decoder, written from the published format descriptions
// (xz-file-format-1.0.4, the LZMA SDK "lzma-specification").  The reference gets this from the
// un-vendored crate xz2 0.1 = liblzma (lib/src/parser/v1.rs:19, :59-67); the build image has the
// liblzma runtime but no lzma.h, so the product carries its own decoder.
// Single-shot: the whole compressed chunk is in memory, the whole output is produced at once.
#include "xz_dec.h"

#include <algorithm>
#include <cstring>

namespace glz {
namespace {

// ---- CRC32 (IEEE) and CRC64 (ECMA-182), both reflected --------------------------------------
struct CrcTables {
  uint32_t t32[256];
  uint64_t t64[256];
  CrcTables() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      uint64_t d = i;
      for (int k = 0; k < 8; ++k) {
        c = (c & 1) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
        d = (d & 1) ? (d >> 1) ^ 0xC96C5795D7870F42ull : d >> 1;
      }
      t32[i] = c;
      t64[i] = d;
    }
  }
};
const CrcTables& crc_tables() {
  static const CrcTables t;
  return t;
}
uint32_t crc32(const uint8_t* p, size_t n) {
  const auto& t = crc_tables();
  uint32_t c = 0xFFFFFFFFu;
  for (size_t i = 0; i < n; ++i) c = t.t32[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return ~c;
}
uint64_t crc64(const uint8_t* p, size_t n) {
  const auto& t = crc_tables();
  uint64_t c = ~0ull;
  for (size_t i = 0; i < n; ++i) c = t.t64[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return ~c;
}

uint32_t rd32le(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint64_t rd64le(const uint8_t* p) { return (uint64_t)rd32le(p) | ((uint64_t)rd32le(p + 4) << 32); }

// ---- LZMA ---------------------------------------------------------------------------------
constexpr int kNumBitModelTotalBits = 11;
constexpr uint16_t kProbInit = 1 << (kNumBitModelTotalBits - 1);
constexpr int kNumMoveBits = 5;
constexpr uint32_t kTopValue = 1u << 24;

struct RangeDec {
  const uint8_t* p = nullptr;
  const uint8_t* end = nullptr;
  uint32_t range = 0, code = 0;
  bool overrun = false;

  uint8_t next() {
    if (p < end) return *p++;
    overrun = true;
    return 0;
  }
  bool init(const uint8_t* b, const uint8_t* e) {
    p = b;
    end = e;
    overrun = false;
    if (e - b < 5) return false;
    if (next() != 0) return false;
    code = 0;
    for (int i = 0; i < 4; ++i) code = (code << 8) | next();
    range = 0xFFFFFFFFu;
    return true;
  }
  inline void normalize() {
    if (range < kTopValue) {
      range <<= 8;
      code = (code << 8) | next();
    }
  }
  inline unsigned bit(uint16_t& prob) {
    uint32_t bound = (range >> kNumBitModelTotalBits) * prob;
    unsigned b;
    if (code < bound) {
      range = bound;
      prob = (uint16_t)(prob + (((1u << kNumBitModelTotalBits) - prob) >> kNumMoveBits));
      b = 0;
    } else {
      range -= bound;
      code -= bound;
      prob = (uint16_t)(prob - (prob >> kNumMoveBits));
      b = 1;
    }
    normalize();
    return b;
  }
  inline uint32_t direct(int nbits) {
    uint32_t res = 0;
    for (; nbits > 0; --nbits) {
      range >>= 1;
      code -= range;
      uint32_t t = 0u - (code >> 31);
      code += range & t;
      normalize();
      res = (res << 1) + (t + 1);
    }
    return res;
  }
  inline unsigned tree(uint16_t* probs, int nbits) {
    unsigned m = 1;
    for (int i = 0; i < nbits; ++i) m = (m << 1) + bit(probs[m]);
    return m - (1u << nbits);
  }
  inline unsigned tree_rev(uint16_t* probs, int nbits) {
    unsigned m = 1, sym = 0;
    for (int i = 0; i < nbits; ++i) {
      unsigned b = bit(probs[m]);
      m = (m << 1) + b;
      sym |= b << i;
    }
    return sym;
  }
};

struct LenDec {
  uint16_t choice, choice2;
  uint16_t low[16][8], mid[16][8], high[256];
  void reset() {
    choice = choice2 = kProbInit;
    for (auto& r : low)
      for (auto& v : r) v = kProbInit;
    for (auto& r : mid)
      for (auto& v : r) v = kProbInit;
    for (auto& v : high) v = kProbInit;
  }
  unsigned decode(RangeDec& rc, unsigned pos_state) {
    if (rc.bit(choice) == 0) return rc.tree(low[pos_state], 3);
    if (rc.bit(choice2) == 0) return 8 + rc.tree(mid[pos_state], 3);
    return 16 + rc.tree(high, 8);
  }
};

struct Lzma {
  unsigned lc = 0, lp = 0, pb = 0;
  unsigned state = 0;
  uint32_t rep0 = 0, rep1 = 0, rep2 = 0, rep3 = 0;
  uint16_t is_match[12][16], is_rep[12], is_rep_g0[12], is_rep_g1[12], is_rep_g2[12], is_rep0_long[12][16];
  uint16_t pos_slot[4][64], pos_special[115], align[16];
  LenDec len, rep_len;
  std::vector<uint16_t> literal;

  bool set_props(uint8_t d) {
    if (d >= 9 * 5 * 5) return false;
    lc = d % 9;
    d /= 9;
    lp = d % 5;
    pb = d / 5;
    if (lc + lp > 4) return false;  // LZMA2 restriction
    literal.assign((size_t)0x300 << (lc + lp), kProbInit);
    return true;
  }
  void reset_state() {
    state = 0;
    rep0 = rep1 = rep2 = rep3 = 0;
    for (auto& r : is_match)
      for (auto& v : r) v = kProbInit;
    for (auto& r : is_rep0_long)
      for (auto& v : r) v = kProbInit;
    for (int i = 0; i < 12; ++i) is_rep[i] = is_rep_g0[i] = is_rep_g1[i] = is_rep_g2[i] = kProbInit;
    for (auto& r : pos_slot)
      for (auto& v : r) v = kProbInit;
    for (auto& v : pos_special) v = kProbInit;
    for (auto& v : align) v = kProbInit;
    len.reset();
    rep_len.reset();
    for (auto& v : literal) v = kProbInit;
  }
};

// Decodes one LZMA chunk of exactly `unpacked` bytes, appended to out (dictionary = out itself,
// positions counted from dict_start).
bool lzma_chunk(Lzma& s, RangeDec& rc, std::vector<uint8_t>& out, size_t dict_start, size_t unpacked) {
  const size_t target = out.size() + unpacked;
  // geometric growth: an exact-size reserve per (<= 2 MiB) chunk would reallocate and copy the whole output once per chunk
  if (target > out.capacity()) out.reserve(std::max(target, out.capacity() * 2));
  const unsigned pb_mask = (1u << s.pb) - 1, lp_mask = (1u << s.lp) - 1;
  while (out.size() < target) {
    if (rc.overrun) return false;
    const size_t pos = out.size() - dict_start;
    const unsigned pos_state = (unsigned)pos & pb_mask;
    if (rc.bit(s.is_match[s.state][pos_state]) == 0) {
      // literal
      const unsigned prev = pos > 0 ? out.back() : 0;
      uint16_t* probs = &s.literal[(size_t)0x300 * ((((unsigned)pos & lp_mask) << s.lc) + (prev >> (8 - s.lc)))];
      unsigned sym = 1;
      if (s.state >= 7) {
        if (s.rep0 >= pos) return false;
        unsigned match_byte = out[out.size() - s.rep0 - 1];
        do {
          unsigned match_bit = (match_byte >> 7) & 1;
          match_byte <<= 1;
          unsigned b = rc.bit(probs[((1 + match_bit) << 8) + sym]);
          sym = (sym << 1) | b;
          if (match_bit != b) break;
        } while (sym < 0x100);
      }
      while (sym < 0x100) sym = (sym << 1) | rc.bit(probs[sym]);
      out.push_back((uint8_t)sym);
      s.state = s.state < 4 ? 0 : (s.state < 10 ? s.state - 3 : s.state - 6);
      continue;
    }
    unsigned length;
    if (rc.bit(s.is_rep[s.state]) == 0) {
      // simple match
      s.rep3 = s.rep2;
      s.rep2 = s.rep1;
      s.rep1 = s.rep0;
      length = s.len.decode(rc, pos_state);
      s.state = s.state < 7 ? 7 : 10;
      unsigned len_state = length > 3 ? 3 : length;
      unsigned slot = rc.tree(s.pos_slot[len_state], 6);
      uint32_t dist;
      if (slot < 4) {
        dist = slot;
      } else {
        int nbits = (int)(slot >> 1) - 1;
        dist = (2u | (slot & 1)) << nbits;
        if (slot < 14) {
          dist += rc.tree_rev(s.pos_special + dist - slot, nbits);
        } else {
          dist += rc.direct(nbits - 4) << 4;
          dist += rc.tree_rev(s.align, 4);
        }
      }
      s.rep0 = dist;
      if (dist == 0xFFFFFFFFu) return false;  // end marker is not allowed inside LZMA2
    } else {
      if (pos == 0) return false;
      if (rc.bit(s.is_rep_g0[s.state]) == 0) {
        if (rc.bit(s.is_rep0_long[s.state][pos_state]) == 0) {
          // short rep
          s.state = s.state < 7 ? 9 : 11;
          if (s.rep0 >= pos) return false;
          out.push_back(out[out.size() - s.rep0 - 1]);
          continue;
        }
      } else {
        uint32_t dist;
        if (rc.bit(s.is_rep_g1[s.state]) == 0) {
          dist = s.rep1;
        } else {
          if (rc.bit(s.is_rep_g2[s.state]) == 0) {
            dist = s.rep2;
          } else {
            dist = s.rep3;
            s.rep3 = s.rep2;
          }
          s.rep2 = s.rep1;
        }
        s.rep1 = s.rep0;
        s.rep0 = dist;
      }
      length = s.rep_len.decode(rc, pos_state);
      s.state = s.state < 7 ? 8 : 11;
    }
    length += 2;
    if (s.rep0 >= pos) return false;
    if (out.size() + length > target) return false;
    size_t src = out.size() - s.rep0 - 1;
    for (unsigned i = 0; i < length; ++i) out.push_back(out[src + i]);  // may overlap: byte by byte
  }
  return !rc.overrun;
}

bool lzma2_decode(const uint8_t* p, const uint8_t* end, std::vector<uint8_t>& out, const uint8_t** consumed,
                  std::string& err) {
  Lzma s;
  RangeDec rc;
  bool need_dict_reset = true, need_props = true;
  size_t dict_start = out.size();
  for (;;) {
    if (p >= end) { err = "lzma2: truncated stream"; return false; }
    uint8_t ctrl = *p++;
    if (ctrl == 0x00) break;
    if (ctrl == 0x01 || ctrl == 0x02) {
      if (ctrl == 0x01) { need_dict_reset = false; dict_start = out.size(); }
      else if (need_dict_reset) { err = "lzma2: missing dictionary reset"; return false; }
      if (end - p < 2) { err = "lzma2: truncated chunk header"; return false; }
      size_t n = ((size_t)p[0] << 8 | p[1]) + 1;
      p += 2;
      if ((size_t)(end - p) < n) { err = "lzma2: truncated uncompressed chunk"; return false; }
      out.insert(out.end(), p, p + n);
      p += n;
      continue;
    }
    if (ctrl < 0x80) { err = "lzma2: invalid control byte"; return false; }
    if (end - p < 4) { err = "lzma2: truncated chunk header"; return false; }
    size_t unpacked = ((size_t)(ctrl & 0x1F) << 16 | (size_t)p[0] << 8 | p[1]) + 1;
    size_t packed = ((size_t)p[2] << 8 | p[3]) + 1;
    p += 4;
    unsigned reset = (ctrl >> 5) & 3;
    if (reset == 3) { need_dict_reset = false; dict_start = out.size(); }
    else if (need_dict_reset) { err = "lzma2: missing dictionary reset"; return false; }
    if (reset >= 2) {
      if (p >= end) { err = "lzma2: truncated props"; return false; }
      if (!s.set_props(*p++)) { err = "lzma2: bad lc/lp/pb"; return false; }
      need_props = false;
    } else if (need_props) { err = "lzma2: missing properties"; return false; }
    if (reset >= 1) s.reset_state();
    if ((size_t)(end - p) < packed) { err = "lzma2: truncated compressed chunk"; return false; }
    if (!rc.init(p, p + packed)) { err = "lzma2: bad range coder header"; return false; }
    if (!lzma_chunk(s, rc, out, dict_start, unpacked)) { err = "lzma2: corrupt compressed data"; return false; }
    p += packed;
  }
  *consumed = p;
  return true;
}

bool read_vli(const uint8_t*& p, const uint8_t* end, uint64_t& v) {
  v = 0;
  for (int i = 0; i < 9; ++i) {
    if (p >= end) return false;
    uint8_t b = *p++;
    v |= (uint64_t)(b & 0x7F) << (7 * i);
    if (!(b & 0x80)) return !(b == 0 && i > 0);
  }
  return false;
}

}  // namespace

bool xz_decompress(const uint8_t* data, size_t size, std::vector<uint8_t>& out, std::string& err) {
  static const uint8_t kMagic[6] = {0xFD, '7', 'z', 'X', 'Z', 0x00};
  out.clear();
  const uint8_t* p = data;
  const uint8_t* end = data + size;
  bool any_stream = false;
  while (p < end) {
    // stream padding between concatenated streams: multiples of four zero bytes
    if (any_stream) {
      while (end - p >= 4 && rd32le(p) == 0) p += 4;
      if (p == end) break;
    }
    if (end - p < 12 || memcmp(p, kMagic, 6) != 0) { err = "xz: bad stream header magic"; return false; }
    if (p[6] != 0 || (p[7] & 0xF0)) { err = "xz: unsupported stream flags"; return false; }
    if (crc32(p + 6, 2) != rd32le(p + 8)) { err = "xz: stream header CRC mismatch"; return false; }
    const unsigned check_type = p[7] & 0x0F;
    static const unsigned kCheckSize[16] = {0, 4, 4, 4, 8, 8, 8, 16, 16, 16, 32, 32, 32, 64, 64, 64};
    const unsigned check_size = kCheckSize[check_type];
    const uint8_t stream_flags[2] = {p[6], p[7]};
    p += 12;
    any_stream = true;
    // blocks
    for (;;) {
      if (p >= end) { err = "xz: truncated before index"; return false; }
      if (*p == 0x00) break;  // index indicator
      const uint8_t* bh = p;
      size_t hsize = ((size_t)*p + 1) * 4;
      if ((size_t)(end - p) < hsize) { err = "xz: truncated block header"; return false; }
      if (crc32(bh, hsize - 4) != rd32le(bh + hsize - 4)) { err = "xz: block header CRC mismatch"; return false; }
      const uint8_t* q = bh + 1;
      const uint8_t* hend = bh + hsize - 4;
      uint8_t flags = *q++;
      if (flags & 0x3C) { err = "xz: reserved block flags set"; return false; }
      unsigned nfilters = (flags & 3) + 1;
      uint64_t csize = ~0ull, usize = ~0ull;
      if ((flags & 0x40) && !read_vli(q, hend, csize)) { err = "xz: bad compressed size"; return false; }
      if ((flags & 0x80) && !read_vli(q, hend, usize)) { err = "xz: bad uncompressed size"; return false; }
      if (nfilters != 1) { err = "xz: only a single LZMA2 filter is supported"; return false; }
      uint64_t fid, psize;
      if (!read_vli(q, hend, fid) || !read_vli(q, hend, psize)) { err = "xz: bad filter flags"; return false; }
      if (fid != 0x21 || psize != 1 || q >= hend) { err = "xz: unsupported filter"; return false; }
      if ((*q & 0x3F) > 40) { err = "xz: bad LZMA2 dictionary size"; return false; }
      p = bh + hsize;
      const size_t out_before = out.size();
      // the header's uncompressed size (when present and plausible: LZMA2 expands < 2^13 x) sizes the output once
      if (usize != ~0ull && usize <= (uint64_t)(end - p) * 8192u) out.reserve(out.size() + (size_t)usize);
      const uint8_t* cend = nullptr;
      if (!lzma2_decode(p, end, out, &cend, err)) return false;
      if (csize != ~0ull && (uint64_t)(cend - p) != csize) { err = "xz: compressed size mismatch"; return false; }
      if (usize != ~0ull && (uint64_t)(out.size() - out_before) != usize) { err = "xz: uncompressed size mismatch"; return false; }
      size_t pad = (4 - ((size_t)(cend - p) & 3)) & 3;
      p = cend;
      if ((size_t)(end - p) < pad + check_size) { err = "xz: truncated block trailer"; return false; }
      for (size_t i = 0; i < pad; ++i)
        if (p[i] != 0) { err = "xz: nonzero block padding"; return false; }
      p += pad;
      const uint8_t* ud = out.data() + out_before;
      const size_t un = out.size() - out_before;
      if (check_type == 1 && crc32(ud, un) != rd32le(p)) { err = "xz: CRC32 check failed"; return false; }
      if (check_type == 4 && crc64(ud, un) != rd64le(p)) { err = "xz: CRC64 check failed"; return false; }
      p += check_size;  // SHA-256 (10) and reserved checks are skipped, like liblzma without the check compiled in
    }
    // index
    const uint8_t* idx = p;
    ++p;
    uint64_t nrec;
    if (!read_vli(p, end, nrec)) { err = "xz: bad index"; return false; }
    for (uint64_t i = 0; i < nrec; ++i) {
      uint64_t a, b;
      if (!read_vli(p, end, a) || !read_vli(p, end, b)) { err = "xz: bad index record"; return false; }
    }
    while ((size_t)(p - idx) & 3) {
      if (p >= end || *p != 0) { err = "xz: bad index padding"; return false; }
      ++p;
    }
    if (end - p < 4 || crc32(idx, p - idx) != rd32le(p)) { err = "xz: index CRC mismatch"; return false; }
    p += 4;
    const size_t index_size = (size_t)(p - idx);
    // footer
    if (end - p < 12) { err = "xz: truncated stream footer"; return false; }
    if (crc32(p + 4, 6) != rd32le(p)) { err = "xz: stream footer CRC mismatch"; return false; }
    if (((uint64_t)rd32le(p + 4) + 1) * 4 != index_size) { err = "xz: backward size mismatch"; return false; }
    if (p[8] != stream_flags[0] || p[9] != stream_flags[1] || p[10] != 'Y' || p[11] != 'Z') {
      err = "xz: bad stream footer";
      return false;
    }
    p += 12;
  }
  if (!any_stream) { err = "xz: empty input"; return false; }
  return true;
}

}  // namespace glz
