// .xz encoder (single stream, single block, one LZMA2 filter, CRC64 check) for the write side of the
// .glaze format: the reference compresses every chunk with xz2's XzEncoder at preset 9
// (lib/src/parser/v1.rs:49-57).  Any standard-conforming stream is readable by the reference; this one
// is produced by a hash-chain match finder with one-step lazy matching and the usual LZMA models
// (lc = 3, lp = 0, pb = 2), cut into LZMA2 chunks of at most 64 KiB compressed / 2 MiB uncompressed.
#include "xz_enc.h"

#include <algorithm>
#include <cstring>

namespace glz {
namespace {

// ---- checksums ----------------------------------------------------------------------------------
struct Crc {
  uint32_t t32[256];
  uint64_t t64[256];
  Crc() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      uint64_t d = i;
      for (int k = 0; k < 8; ++k) {
        c = (c >> 1) ^ ((c & 1) ? 0xEDB88320u : 0u);
        d = (d >> 1) ^ ((d & 1) ? 0xC96C5795D7870F42ull : 0ull);
      }
      t32[i] = c;
      t64[i] = d;
    }
  }
};
const Crc& crc_tables() {
  static const Crc t;
  return t;
}
uint32_t crc32(const uint8_t* p, size_t n) {
  const Crc& t = crc_tables();
  uint32_t c = 0xFFFFFFFFu;
  for (size_t i = 0; i < n; ++i) c = t.t32[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return ~c;
}
uint64_t crc64(const uint8_t* p, size_t n) {
  const Crc& t = crc_tables();
  uint64_t c = ~0ull;
  for (size_t i = 0; i < n; ++i) c = t.t64[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return ~c;
}
void put32(std::vector<uint8_t>& o, uint32_t v) { for (int k = 0; k < 4; ++k) o.push_back((uint8_t)(v >> (8 * k))); }
void put64(std::vector<uint8_t>& o, uint64_t v) { for (int k = 0; k < 8; ++k) o.push_back((uint8_t)(v >> (8 * k))); }
void put_varint(std::vector<uint8_t>& o, uint64_t v) {
  while (v >= 0x80) { o.push_back((uint8_t)(v | 0x80)); v >>= 7; }
  o.push_back((uint8_t)v);
}

// ---- range coder --------------------------------------------------------------------------------
typedef uint16_t Prob;
constexpr int kBitModelTotalBits = 11, kMoveBits = 5;
constexpr Prob kProbInit = 1 << (kBitModelTotalBits - 1);

struct RangeEnc {
  std::vector<uint8_t> out;
  uint64_t low = 0;
  uint32_t range = 0xFFFFFFFFu;
  uint8_t cache = 0;
  uint64_t cache_size = 1;
  void reset() { out.clear(); low = 0; range = 0xFFFFFFFFu; cache = 0; cache_size = 1; }
  void shift_low() {
    if ((uint32_t)low < 0xFF000000u || (low >> 32) != 0) {
      const uint8_t carry = (uint8_t)(low >> 32);
      uint8_t c = cache;
      do {
        out.push_back((uint8_t)(c + carry));
        c = 0xFF;
      } while (--cache_size != 0);
      cache = (uint8_t)(low >> 24);
    }
    ++cache_size;
    low = (low & 0x00FFFFFFull) << 8;
  }
  void bit(Prob& p, uint32_t b) {
    const uint32_t bound = (range >> kBitModelTotalBits) * p;
    if (!b) {
      range = bound;
      p = (Prob)(p + (((1u << kBitModelTotalBits) - p) >> kMoveBits));
    } else {
      low += bound;
      range -= bound;
      p = (Prob)(p - (p >> kMoveBits));
    }
    while (range < (1u << 24)) { range <<= 8; shift_low(); }
  }
  void direct(uint32_t value, int nbits) {
    for (int i = nbits - 1; i >= 0; --i) {
      range >>= 1;
      if ((value >> i) & 1) low += range;
      while (range < (1u << 24)) { range <<= 8; shift_low(); }
    }
  }
  void flush() { for (int i = 0; i < 5; ++i) shift_low(); }
  size_t pending() const { return out.size() + (size_t)cache_size + 4; }   // upper bound of the size after flush()
};

void bittree(RangeEnc& rc, Prob* probs, int nbits, uint32_t sym) {
  uint32_t m = 1;
  for (int i = nbits - 1; i >= 0; --i) {
    const uint32_t b = (sym >> i) & 1;
    rc.bit(probs[m], b);
    m = (m << 1) | b;
  }
}
void bittree_reverse(RangeEnc& rc, Prob* probs, int nbits, uint32_t sym) {
  uint32_t m = 1;
  for (int i = 0; i < nbits; ++i) {
    const uint32_t b = sym & 1;
    sym >>= 1;
    rc.bit(probs[m], b);
    m = (m << 1) | b;
  }
}

// ---- LZMA models --------------------------------------------------------------------------------
constexpr int kLc = 3, kLp = 0, kPb = 2;
constexpr int kNumStates = 12, kPosStates = 1 << kPb;
constexpr int kMatchMin = 2, kMatchMax = 273;
constexpr int kNumLenToPosStates = 4, kEndPosModelIndex = 14, kNumFullDistances = 1 << (kEndPosModelIndex / 2), kAlignBits = 4;

struct LenEnc {
  Prob choice, choice2, low[kPosStates][8], mid[kPosStates][8], high[256];
  void init() {
    choice = choice2 = kProbInit;
    for (auto& a : low) for (auto& p : a) p = kProbInit;
    for (auto& a : mid) for (auto& p : a) p = kProbInit;
    for (auto& p : high) p = kProbInit;
  }
  void encode(RangeEnc& rc, uint32_t len, uint32_t pos_state) {   // len is the real length (>= 2)
    len -= kMatchMin;
    if (len < 8) {
      rc.bit(choice, 0);
      bittree(rc, low[pos_state], 3, len);
    } else if (len < 16) {
      rc.bit(choice, 1);
      rc.bit(choice2, 0);
      bittree(rc, mid[pos_state], 3, len - 8);
    } else {
      rc.bit(choice, 1);
      rc.bit(choice2, 1);
      bittree(rc, high, 8, len - 16);
    }
  }
};

struct Model {
  Prob is_match[kNumStates][kPosStates], is_rep[kNumStates], is_rep0[kNumStates], is_rep1[kNumStates], is_rep2[kNumStates];
  Prob is_rep0_long[kNumStates][kPosStates];
  Prob literal[(size_t)0x300 << (kLc + kLp)];
  Prob pos_slot[kNumLenToPosStates][64];
  Prob pos_special[kNumFullDistances - kEndPosModelIndex + 1];
  Prob align[1 << kAlignBits];
  LenEnc match_len, rep_len;
  uint32_t state = 0;
  uint32_t rep[4] = {0, 0, 0, 0};   // distances - 1
  void init() {
    auto fill = [](Prob* p, size_t n) { for (size_t i = 0; i < n; ++i) p[i] = kProbInit; };
    fill(&is_match[0][0], sizeof(is_match) / sizeof(Prob));
    fill(is_rep, kNumStates); fill(is_rep0, kNumStates); fill(is_rep1, kNumStates); fill(is_rep2, kNumStates);
    fill(&is_rep0_long[0][0], sizeof(is_rep0_long) / sizeof(Prob));
    fill(literal, sizeof(literal) / sizeof(Prob));
    fill(&pos_slot[0][0], sizeof(pos_slot) / sizeof(Prob));
    fill(pos_special, sizeof(pos_special) / sizeof(Prob));
    fill(align, sizeof(align) / sizeof(Prob));
    match_len.init();
    rep_len.init();
    state = 0;
    rep[0] = rep[1] = rep[2] = rep[3] = 0;
  }
};

inline uint32_t pos_slot_of(uint32_t dist) {   // dist = distance - 1
  if (dist < 4) return dist;
  const int n = 31 - __builtin_clz(dist);
  return (uint32_t)(2 * n) + ((dist >> (n - 1)) & 1);
}

struct Encoder {
  const uint8_t* in;
  size_t size;
  Model m;
  RangeEnc rc;

  void literal(size_t pos) {
    const uint32_t pos_state = (uint32_t)pos & (kPosStates - 1);
    rc.bit(m.is_match[m.state][pos_state], 0);
    const uint8_t prev = pos ? in[pos - 1] : 0;
    Prob* probs = m.literal + (size_t)0x300 * (prev >> (8 - kLc));   // lp = 0
    const uint32_t sym = in[pos];
    if (m.state < 7) {
      bittree(rc, probs, 8, sym);
    } else {
      uint32_t match_byte = in[pos - m.rep[0] - 1];
      uint32_t offs = 0x100, s = 1;
      for (int i = 7; i >= 0; --i) {
        match_byte <<= 1;
        const uint32_t mb = match_byte & offs;
        const uint32_t b = (sym >> i) & 1;
        rc.bit(probs[offs + mb + s], b);
        s = (s << 1) | b;
        offs &= b ? mb : ~mb;
      }
    }
    m.state = m.state < 4 ? 0 : (m.state < 10 ? m.state - 3 : m.state - 6);
  }
  void match(size_t pos, uint32_t dist, uint32_t len) {   // dist = distance - 1
    const uint32_t pos_state = (uint32_t)pos & (kPosStates - 1);
    rc.bit(m.is_match[m.state][pos_state], 1);
    rc.bit(m.is_rep[m.state], 0);
    m.match_len.encode(rc, len, pos_state);
    const uint32_t len_state = std::min<uint32_t>(len - kMatchMin, kNumLenToPosStates - 1);
    const uint32_t slot = pos_slot_of(dist);
    bittree(rc, m.pos_slot[len_state], 6, slot);
    if (slot >= 4) {
      const int footer = (int)(slot >> 1) - 1;
      const uint32_t base = (2u | (slot & 1u)) << footer;
      const uint32_t reduced = dist - base;
      if (slot < (uint32_t)kEndPosModelIndex) {
        bittree_reverse(rc, m.pos_special + base - slot - 1, footer, reduced);
      } else {
        rc.direct(reduced >> kAlignBits, footer - kAlignBits);
        bittree_reverse(rc, m.align, kAlignBits, reduced & ((1u << kAlignBits) - 1));
      }
    }
    m.rep[3] = m.rep[2]; m.rep[2] = m.rep[1]; m.rep[1] = m.rep[0]; m.rep[0] = dist;
    m.state = m.state < 7 ? 7 : 10;
  }
  void rep_match(size_t pos, int idx, uint32_t len) {
    const uint32_t pos_state = (uint32_t)pos & (kPosStates - 1);
    rc.bit(m.is_match[m.state][pos_state], 1);
    rc.bit(m.is_rep[m.state], 1);
    if (idx == 0) {
      rc.bit(m.is_rep0[m.state], 0);
      rc.bit(m.is_rep0_long[m.state][pos_state], len == 1 ? 0 : 1);
    } else {
      const uint32_t d = m.rep[idx];
      rc.bit(m.is_rep0[m.state], 1);
      if (idx == 1) {
        rc.bit(m.is_rep1[m.state], 0);
      } else {
        rc.bit(m.is_rep1[m.state], 1);
        rc.bit(m.is_rep2[m.state], idx == 2 ? 0 : 1);
        if (idx == 3) m.rep[3] = m.rep[2];
        m.rep[2] = m.rep[1];
      }
      m.rep[1] = m.rep[0];
      m.rep[0] = d;
    }
    if (len == 1) {
      m.state = m.state < 7 ? 9 : 11;
    } else {
      m.rep_len.encode(rc, len, pos_state);
      m.state = m.state < 7 ? 8 : 11;
    }
  }
};

// ---- match finder: hash chains over 4-byte (and 3-byte) prefixes ---------------------------------
struct MatchFinder {
  const uint8_t* in;
  size_t size;
  uint32_t window;            // maximum distance
  std::vector<uint32_t> head4, head3, prev;
  static constexpr uint32_t kNil = 0xFFFFFFFFu;
  static constexpr int kHashBits4 = 20, kHashBits3 = 16, kDepth = 64, kNiceLen = 128;
  MatchFinder(const uint8_t* d, size_t n, uint32_t w) : in(d), size(n), window(w), head4(1u << kHashBits4, kNil), head3(1u << kHashBits3, kNil), prev(n, kNil) {}
  static uint32_t h4(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return (v * 2654435761u) >> (32 - kHashBits4); }
  static uint32_t h3(const uint8_t* p) { const uint32_t v = p[0] | (p[1] << 8) | (p[2] << 16); return (v * 2654435761u) >> (32 - kHashBits3); }
  void insert(size_t pos) {
    if (pos + 4 <= size) {
      const uint32_t h = h4(in + pos);
      prev[pos] = head4[h];
      head4[h] = (uint32_t)pos;
    }
    if (pos + 3 <= size) head3[h3(in + pos)] = (uint32_t)pos;
  }
  uint32_t match_len(size_t a, size_t b, uint32_t limit) const {   // a < b
    uint32_t l = 0;
    while (l < limit && in[a + l] == in[b + l]) ++l;
    return l;
  }
  // longest match at pos (not yet inserted); returns length (0 = none) and distance - 1
  uint32_t find(size_t pos, uint32_t& dist) const {
    const uint32_t limit = (uint32_t)std::min<size_t>(kMatchMax, size - pos);
    uint32_t best = 0;
    if (limit < 3) return 0;
    if (limit >= 4) {
      uint32_t cand = head4[h4(in + pos)];
      for (int depth = 0; cand != kNil && depth < kDepth; ++depth) {
        if (pos - cand > window) break;
        if (in[cand + best] == in[pos + best]) {
          const uint32_t l = match_len(cand, pos, limit);
          if (l > best) {
            best = l;
            dist = (uint32_t)(pos - cand - 1);
            if (l >= (uint32_t)kNiceLen || l == limit) break;
          }
        }
        cand = prev[cand];
      }
    }
    if (best < 3) {
      const uint32_t cand = head3[h3(in + pos)];
      if (cand != kNil && pos - cand <= window && pos - cand <= (1u << 14)) {   // a 3-byte match only pays at short range
        const uint32_t l = match_len(cand, pos, limit);
        if (l >= 3 && l > best) { best = l; dist = (uint32_t)(pos - cand - 1); }
      }
    }
    return best >= 3 ? best : 0;
  }
};

uint8_t dict_size_code(uint64_t want) {   // smallest LZMA2 dictionary code whose size is >= want (at least 4 KiB)
  for (uint8_t b = 0; b < 40; ++b) {
    const uint64_t sz = (uint64_t)(2 | (b & 1)) << (b / 2 + 11);
    if (sz >= want) return b;
  }
  return 40;
}

}  // namespace

bool xz_compress(const uint8_t* data, size_t size, std::vector<uint8_t>& out) {
  out.clear();
  if (size >= 0xFFFFFF00ull) return false;   // positions are 32-bit in the match finder
  std::vector<uint8_t> lzma2;
  const uint64_t kMaxDict = 64ull << 20;   // preset 9's dictionary
  const uint8_t dict_code = dict_size_code(std::min<uint64_t>(std::max<uint64_t>(size, 4096), kMaxDict));
  const uint64_t dict_size = (uint64_t)(2 | (dict_code & 1)) << (dict_code / 2 + 11);
  if (size > 0) {
    Encoder enc;
    enc.in = data;
    enc.size = size;
    enc.m.init();
    MatchFinder mf(data, size, (uint32_t)std::min<uint64_t>(dict_size, 0xFFFFFFFFull) - 1);
    size_t pos = 0;
    bool first = true;
    while (pos < size) {
      // one LZMA2 chunk
      enc.rc.reset();
      const size_t chunk_start = pos;
      constexpr size_t kMaxUnpacked = (2u << 20) - kMatchMax - 1, kMaxPacked = 65536 - 64;
      while (pos < size && pos - chunk_start < kMaxUnpacked && enc.rc.pending() < kMaxPacked) {
        // candidates: repeated distances first (cheapest to code), then the hash chains
        const uint32_t limit = (uint32_t)std::min<size_t>(kMatchMax, size - pos);
        uint32_t rep_len = 0;
        int rep_idx = -1;
        for (int r = 0; r < 4; ++r) {
          const size_t d = (size_t)enc.m.rep[r] + 1;
          if (d > pos) continue;
          const uint32_t l = mf.match_len(pos - d, pos, limit);
          if (l >= 2 && l > rep_len) { rep_len = l; rep_idx = r; }
        }
        uint32_t dist = 0;
        uint32_t len = mf.find(pos, dist);
        if (len == 3 && dist >= (1u << 14)) len = 0;
        // one-step lazy evaluation: a clearly longer match starting at the next byte wins over a short one here
        if (len >= 3 && len < 32 && rep_len + 1 < len && pos + 1 < size) {
          mf.insert(pos);
          uint32_t dist2 = 0;
          const uint32_t len2 = mf.find(pos + 1, dist2);
          if (len2 > len + 1) {
            enc.literal(pos);
            ++pos;
            continue;   // the next iteration finds (at least) that match again
          }
          // fall through: code the match found at pos (pos is already inserted)
          if (rep_len >= 2 && rep_len + 2 >= len) {
            enc.rep_match(pos, rep_idx, rep_len);
            for (size_t k = 1; k < rep_len; ++k) mf.insert(pos + k);
            pos += rep_len;
          } else {
            enc.match(pos, dist, len);
            for (size_t k = 1; k < len; ++k) mf.insert(pos + k);
            pos += len;
          }
          continue;
        }
        if (rep_len >= 2 && (rep_len + 2 >= len || len == 0)) {
          enc.rep_match(pos, rep_idx, rep_len);
          for (size_t k = 0; k < rep_len; ++k) mf.insert(pos + k);
          pos += rep_len;
        } else if (len >= 3) {
          enc.match(pos, dist, len);
          for (size_t k = 0; k < len; ++k) mf.insert(pos + k);
          pos += len;
        } else if (pos > 0 && (size_t)enc.m.rep[0] + 1 <= pos && data[pos] == data[pos - enc.m.rep[0] - 1]) {
          enc.rep_match(pos, 0, 1);   // short rep: one byte at the last distance
          mf.insert(pos);
          ++pos;
        } else {
          enc.literal(pos);
          mf.insert(pos);
          ++pos;
        }
      }
      enc.rc.flush();
      const size_t unpacked = pos - chunk_start, packed = enc.rc.out.size();
      // control: 0xE0 = LZMA, dictionary + state reset + new properties (first chunk); 0x80 = LZMA, nothing reset
      lzma2.push_back((uint8_t)((first ? 0xE0 : 0x80) | (((unpacked - 1) >> 16) & 0x1F)));
      lzma2.push_back((uint8_t)((unpacked - 1) >> 8));
      lzma2.push_back((uint8_t)(unpacked - 1));
      lzma2.push_back((uint8_t)((packed - 1) >> 8));
      lzma2.push_back((uint8_t)(packed - 1));
      if (first) lzma2.push_back((uint8_t)((kPb * 5 + kLp) * 9 + kLc));
      lzma2.insert(lzma2.end(), enc.rc.out.begin(), enc.rc.out.end());
      first = false;
    }
  }
  lzma2.push_back(0x00);   // end of LZMA2 data

  // stream header
  const uint8_t magic[6] = {0xFD, '7', 'z', 'X', 'Z', 0x00};
  const uint8_t flags[2] = {0x00, 0x04};   // check = CRC64
  out.insert(out.end(), magic, magic + 6);
  out.insert(out.end(), flags, flags + 2);
  put32(out, crc32(flags, 2));
  uint64_t unpadded = 0;
  if (size > 0) {
  // block header: size, flags (one filter, no size fields), LZMA2 filter id, props size, dictionary size, padding, CRC32
  const size_t block_start = out.size();
  const uint8_t bh[8] = {0x02, 0x00, 0x21, 0x01, dict_code, 0x00, 0x00, 0x00};   // (0x02 + 1) * 4 = 12 bytes with the CRC
  out.insert(out.end(), bh, bh + 8);
  put32(out, crc32(bh, 8));
  const size_t header_size = out.size() - block_start;
  out.insert(out.end(), lzma2.begin(), lzma2.end());
  unpadded = header_size + lzma2.size() + 8;
  while ((out.size() - block_start) % 4) out.push_back(0);
  put64(out, crc64(data, size));
  }
  // index
  std::vector<uint8_t> index;
  index.push_back(0x00);
  put_varint(index, size > 0 ? 1 : 0);   // empty input: a stream without blocks, like xz itself writes
  if (size > 0) {
    put_varint(index, unpadded);
    put_varint(index, size);
  }
  while (index.size() % 4) index.push_back(0);
  put32(index, crc32(index.data(), index.size()));
  out.insert(out.end(), index.begin(), index.end());
  // footer: CRC32 of (backward size, flags), backward size, flags, magic
  std::vector<uint8_t> tail;
  put32(tail, (uint32_t)(index.size() / 4 - 1));
  tail.push_back(0x00);
  tail.push_back(0x04);
  put32(out, crc32(tail.data(), tail.size()));
  out.insert(out.end(), tail.begin(), tail.end());
  out.push_back('Y');
  out.push_back('Z');
  return true;
}

}  // namespace glz
