// pdlp_mps.cpp — see pdlp_mps.hpp.  Reference behaviour followed: io/HMpsFF.cpp (free-format MPS parser).
#include "pdlp_mps.hpp"
#include "pdlp_env.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <exception>
#include <mutex>
#include <thread>
#include <unordered_map>

namespace pdlp {
namespace mps {
namespace {

constexpr double kInf = std::numeric_limits<double>::infinity();

// ---- views, words, numbers ----------------------------------------------------------------------------------
struct Sv {
  const char* p = nullptr;
  uint32_t n = 0;
  bool empty() const { return n == 0; }
  bool is(const char* s) const { return n == std::strlen(s) && std::memcmp(p, s, n) == 0; }
  bool operator==(const Sv& o) const { return n == o.n && std::memcmp(p, o.p, n) == 0; }
  std::string str() const { return std::string(p, n); }
};

// the reference's word separators (util/stringutil.h:31 default_non_chars)
inline bool isWs(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; }

inline Sv nextWord(const char*& p, const char* e) {
  while (p < e && isWs(*p)) ++p;
  const char* s = p;
  while (p < e && !isWs(*p)) ++p;
  return Sv{s, (uint32_t)(p - s)};
}
inline bool restIsBlank(const char* p, const char* e) {
  while (p < e && isWs(*p)) ++p;
  return p == e;
}

// HMpsFF::getValue (HMpsFF.cpp:2088-2112): the first 'D' (else the first 'd') becomes 'E', then atof.
// Fast path: std::from_chars rounds exactly as strtod does; anything it does not consume completely (exponent
// letter D, a sign it does not take, hex, trailing text, out of range) goes through the literal rule.
double parseValue(Sv w) {
  if (w.n == 0) return 0.0;
  const char* p = w.p;
  const char* e = w.p + w.n;
  if (*p == '+' && w.n > 1 && ((p[1] >= '0' && p[1] <= '9') || p[1] == '.')) ++p;
  double v;
  const auto r = std::from_chars(p, e, v);
  if (r.ec == std::errc() && r.ptr == e) return v;
  char buf[128];
  std::string big;
  char* s = buf;
  if (w.n >= sizeof(buf)) { big.assign(w.p, w.n); s = &big[0]; }
  else { std::memcpy(buf, w.p, w.n); buf[w.n] = 0; }
  char* d = std::strchr(s, 'D');
  if (!d) d = std::strchr(s, 'd');
  if (d) *d = 'E';
  return std::atof(s);
}

// ---- lines ------------------------------------------------------------------------------------------------------
// One line of the mapping, already trimmed as HMpsFF::getMpsLine does (:222-244); skip = blank line or a '*' in
// the first column of the untrimmed line.
struct Line {
  const char* b;
  const char* e;
  bool skip;
};
inline const char* lineEnd(const char* p, const char* fileEnd) {
  const void* q = std::memchr(p, '\n', (size_t)(fileEnd - p));
  return q ? (const char*)q : fileEnd;
}
inline Line trimLine(const char* b, const char* nl) {
  Line L{b, nl, false};
  if (b == nl || *b == '*') { L.skip = true; return L; }
  while (L.b < L.e && isWs(*L.b)) ++L.b;
  while (L.e > L.b && isWs(L.e[-1])) --L.e;
  L.skip = L.b == L.e;
  return L;
}
// first line start at or after pos inside [lo, hi)
inline const char* alignToLine(const char* lo, const char* hi, const char* pos) {
  if (pos <= lo) return lo;
  if (pos >= hi) return hi;
  const void* q = std::memchr(pos - 1, '\n', (size_t)(hi - (pos - 1)));
  return q ? (const char*)q + 1 : hi;
}

// An exception inside a worker (bad_alloc from a table, a Fail from the parser) must not reach std::terminate — the
// host process is HiGHS: the first one is kept, every thread is joined, then it is thrown again on the calling thread.
template <class Fn>
void parallelFor(int T, Fn&& fn) {
  if (T <= 1) { fn(0); return; }
  std::exception_ptr first;
  std::mutex mu;
  auto guarded = [&](int t) {
    try {
      fn(t);
    } catch (...) {
      std::lock_guard<std::mutex> lock(mu);
      if (!first) first = std::current_exception();
    }
  };
  std::vector<std::thread> th;
  th.reserve((size_t)T - 1);
  int spawned = 1;  // pieces 1 .. spawned-1 have a thread of their own
  try {
    for (; spawned < T; ++spawned) th.emplace_back(guarded, spawned);
  } catch (...) {
    // thread creation failed (EAGAIN under a process limit): not an error of the read — the pieces that got no thread
    // run on the calling thread below
  }
  guarded(0);
  for (int t = spawned; t < T; ++t) guarded(t);
  for (auto& x : th) x.join();
  if (first) std::rethrow_exception(first);
}
// Zero-filled array for the randomly accessed tables (name slots, per-thread row stamps): anonymous mapping with
// transparent huge pages requested, so that a lookup in a 64 MB table does not also miss the TLB.
template <class E>
struct BigArray {
  E* p = nullptr;
  size_t n = 0, bytes = 0;
  BigArray() = default;
  BigArray(const BigArray&) = delete;
  BigArray& operator=(const BigArray&) = delete;
  ~BigArray() { release(); }
  void release() {
    if (p) ::munmap((void*)p, bytes);
    p = nullptr;
    n = bytes = 0;
  }
  void allocZero(size_t count) {
    release();
    n = count;
    bytes = ((count ? count : 1) * sizeof(E) + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
    void* m = ::mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) throw std::bad_alloc();
    ::madvise(m, bytes, MADV_HUGEPAGE);
    p = (E*)m;
  }
  E& operator[](size_t i) { return p[i]; }
  const E& operator[](size_t i) const { return p[i]; }
};

// piece t of T of the byte range [lo, hi), cut at line starts
inline void piece(const char* lo, const char* hi, int t, int T, const char*& b, const char*& e) {
  const int64_t len = hi - lo;
  b = alignToLine(lo, hi, lo + len * t / T);
  e = alignToLine(lo, hi, lo + len * (t + 1) / T);
}

// ---- section keywords (HMpsFF::checkFirstWord, :399-489) --------------------------------------------------------
enum Key : uint8_t {
  kNone, kName, kObjsense, kMax, kMin, kRows, kCols, kRhs, kBounds, kRanges, kQsection, kQmatrix, kQuadobj, kQcmatrix,
  kCsection, kDelayedrows, kModelcuts, kUsercuts, kIndicators, kSets, kSos, kGencons, kPwlobj, kPwlnam, kPwlcon, kEnd
};
Key keyOfLine(const Line& L, const char*& afterWord) {
  const char* p = L.b;
  const Sv w = nextWord(p, L.e);
  afterWord = p;
  if (w.n < 3) return kNone;  // a single character is never a keyword (:401-405); no keyword has two letters
  // keywords have 3, 4 or 6..11 letters; any word starting with MAX / MIN counts as that key
  const char c0 = (char)(w.p[0] & ~0x20), c1 = (char)(w.p[1] & ~0x20), c2 = (char)(w.p[2] & ~0x20);
  const bool maxMin = c0 == 'M' && ((c1 == 'A' && c2 == 'X') || (c1 == 'I' && c2 == 'N'));
  if (!maxMin && (w.n == 5 || w.n > 11)) return kNone;
  char u[12];
  const uint32_t k = w.n < 11 ? w.n : 11;
  for (uint32_t i = 0; i < k; ++i) {
    const char c = w.p[i];
    u[i] = (c >= 'a' && c <= 'z') ? (char)(c - 32) : c;
  }
  u[k] = 0;
  Key key = kNone;
  if (maxMin) {
    // NAME / OBJSENSE / ... cannot start with MAX or MIN, so the prefix rule (checked second in the reference) decides
    key = c1 == 'A' ? kMax : kMin;
  } else {
    struct Kw { const char* name; uint8_t len; Key key; };
    static const Kw kKeywords[] = {
        {"NAME", 4, kName}, {"OBJSENSE", 8, kObjsense}, {"ROWS", 4, kRows}, {"COLUMNS", 7, kCols}, {"RHS", 3, kRhs},
        {"BOUNDS", 6, kBounds}, {"RANGES", 6, kRanges}, {"QSECTION", 8, kQsection}, {"QMATRIX", 7, kQmatrix},
        {"QUADOBJ", 7, kQuadobj}, {"QCMATRIX", 8, kQcmatrix}, {"CSECTION", 8, kCsection}, {"DELAYEDROWS", 11, kDelayedrows},
        {"MODELCUTS", 9, kModelcuts}, {"USERCUTS", 8, kUsercuts}, {"INDICATORS", 10, kIndicators}, {"SETS", 4, kSets},
        {"SOS", 3, kSos}, {"GENCONS", 7, kGencons}, {"PWLOBJ", 6, kPwlobj}, {"PWLNAM", 6, kPwlnam}, {"PWLCON", 6, kPwlcon},
        {"ENDATA", 6, kEnd}};
    for (const Kw& kw : kKeywords)
      if (kw.len == w.n && kw.name[0] == u[0] && std::memcmp(u, kw.name, w.n) == 0) { key = kw.key; break; }
    if (key == kNone) return kNone;
  }
  // keywords may be column / RHS / bound-set names: they open a section only when alone on the line, except the
  // five that take arguments (:479-488)
  if (key == kName || key == kObjsense || key == kQcmatrix || key == kQsection || key == kCsection) return key;
  return restIsBlank(p, L.e) ? key : kNone;
}

// ---- name tables ------------------------------------------------------------------------------------------------
inline uint64_t hashName(const char* p, uint32_t n) {
  uint64_t h = 0x9E3779B97F4A7C15ull ^ ((uint64_t)n << 56);
  while (n >= 8) {
    uint64_t w;
    std::memcpy(&w, p, 8);
    h = (h ^ w) * 0xff51afd7ed558ccdull;
    h ^= h >> 32;
    p += 8;
    n -= 8;
  }
  uint64_t w = 0;
  std::memcpy(&w, p, n);
  h = (h ^ w) * 0xc4ceb9fe1a85ec53ull;
  h ^= h >> 29;
  h *= 0x9E3779B97F4A7C15ull;
  h ^= h >> 32;
  return h;
}

// name -> value, FIRST insertion wins (the reference's unordered_map::emplace).  64 shards by the top hash bits,
// so the table is built by all threads at once (each thread owns whole shards and scans the names in file order)
// and is read-only afterwards.  A slot carries the full hash, the value and the name itself when it has at most
// 16 characters, so a lookup costs one cache miss (two for longer names) however large the model is.
class NameTable {
 public:
  static constexpr int kShards = 64;
  void build(const std::vector<Sv>& names, const std::vector<int32_t>& values, int T, std::vector<uint8_t>* duplicate) {
    const size_t n = names.size();
    std::vector<uint64_t> hash(n);
    std::vector<uint32_t> hist((size_t)T * kShards, 0);
    parallelFor(T, [&](int t) {
      uint32_t* h = &hist[(size_t)t * kShards];
      for (size_t i = n * t / T; i < n * (t + 1) / T; ++i) {
        hash[i] = hashName(names[i].p, names[i].n);
        ++h[hash[i] >> 58];
      }
    });
    if (duplicate) duplicate->assign(n, 0);
    size_t total = 0;
    for (int s = 0; s < kShards; ++s) {
      uint32_t count = 0;
      for (int u = 0; u < T; ++u) count += hist[(size_t)u * kShards + s];
      uint32_t cap = 16;
      while (cap < 2 * count + 2) cap <<= 1;
      shardOff_[s] = total;
      shardMask_[s] = cap - 1;
      total += cap;
    }
    slots_.allocZero(total);  // zero = empty slot (a name has at least one character)
    parallelFor(T, [&](int t) {
      for (size_t i = 0; i < n; ++i) {
        const uint64_t h = hash[i];
        const int s = (int)(h >> 58);
        if (s % T != t) continue;
        Slot* slot = slots_.p + shardOff_[s];
        const uint32_t mask = shardMask_[s];
        uint32_t k = (uint32_t)h & mask;
        for (;;) {
          Slot& q = slot[k];
          if (q.len == 0) {
            q.hash = h;
            q.value = values[i];
            q.len = names[i].n;
            if (names[i].n <= sizeof(q.in)) std::memcpy(q.in, names[i].p, names[i].n);
            else q.p = names[i].p;
            break;
          }
          if (q.hash == h && q.equals(names[i])) { if (duplicate) (*duplicate)[i] = 1; break; }
          k = (k + 1) & mask;
        }
      }
    });
    built_ = n > 0;
  }
  bool find(Sv w, int32_t& value) const {
    if (!built_) return false;
    const uint64_t h = hashName(w.p, w.n);
    const Slot* slot = slots_.p + shardOff_[h >> 58];
    const uint32_t mask = shardMask_[h >> 58];
    uint32_t k = (uint32_t)h & mask;
    for (;;) {
      const Slot& q = slot[k];
      if (q.len == 0) return false;
      if (q.hash == h && q.equals(w)) { value = q.value; return true; }
      k = (k + 1) & mask;
    }
  }

 private:
  struct Slot {  // 32 bytes; all-zero = empty
    uint64_t hash;
    int32_t value;
    uint32_t len;
    union {
      char in[16];
      const char* p;
    };
    bool equals(const Sv& w) const { return len == w.n && std::memcmp(len <= sizeof(in) ? in : p, w.p, w.n) == 0; }
  };
  bool built_ = false;
  BigArray<Slot> slots_;
  size_t shardOff_[kShards] = {};
  uint32_t shardMask_[kShards] = {};
};

constexpr int32_t kObjRow = -1, kFreeRow = -2;  // rowname2idx values of the cost row and of the other N rows (:667-676)

struct Fail {
  ReadStatus status;
  std::string msg;
};

// ---- COLUMNS (HMpsFF::parseCols, :715-1042) ---------------------------------------------------------------------
struct ColPiece {
  std::vector<Sv> runName;        // one run = consecutive lines with the same first word
  std::vector<int64_t> runBeg;    // entries of run r: [runBeg[r], runBeg[r+1])
  std::vector<double> runCost;
  int64_t firstKept = -1;         // >= 0: run 0 continues the previous piece's last column with this many entries
  std::vector<int32_t> row;
  std::vector<double> val;
  struct Marker { uint32_t runsBefore; uint8_t kind; };  // kind: 0 'INTORG', 1 'INTEND', 2 anything else
  std::vector<Marker> markers;
  uint64_t ignoredRow = 0, dupCost = 0, dupNz = 0;
  Sv firstIgnoredRow;
  bool failed = false;
  Fail fail{kReadError, ""};
};

void parseColumnsPiece(const char* b, const char* e, const char* fileEnd, const NameTable& rows, int32_t numRow, ColPiece& P) {
  BigArray<int32_t> stamp;  // stamp[row] = 1 + the run that last used the row
  stamp.allocZero((size_t)std::max(numRow, 1));
  const size_t guess = (size_t)(e - b) / 24 + 16;
  P.row.reserve(guess);
  P.val.reserve(guess);
  int32_t runId = -1;
  Sv cur;
  double cost = 0.0;
  auto closeRun = [&] {
    if (runId >= 0) P.runCost.push_back(cost);
  };
  auto entry = [&](int32_t idx, double value) {
    if (value == 0.0) return;  // zeros are dropped before any other rule (:897); a NaN is kept, as atof gives it
    if (idx >= 0) {
      if (stamp[(size_t)idx] == runId + 1) { ++P.dupNz; return; }  // first value of a (column, row) pair wins (:900-912)
      stamp[(size_t)idx] = runId + 1;
      P.row.push_back(idx);
      P.val.push_back(value);
    } else if (idx == kObjRow) {
      if (cost != 0.0 || cost != cost) ++P.dupCost;  // `if (col_cost)`: nonzero, NaN included
      else cost = value;
    }
  };
  for (const char* p = b; p < e;) {
    const char* nl = lineEnd(p, fileEnd);
    const Line L = trimLine(p, nl);
    p = nl + 1;
    if (L.skip) continue;
    const char* q = L.b;
    const Sv w0 = nextWord(q, L.e);
    const Sv w1 = nextWord(q, L.e);
    if (w1.is("'MARKER'")) {
      const Sv w2 = nextWord(q, L.e);
      P.markers.push_back({(uint32_t)P.runName.size(), (uint8_t)(w2.is("'INTORG'") ? 0 : w2.is("'INTEND'") ? 1 : 2)});
      continue;
    }
    int32_t idx1 = 0;
    const bool found1 = !w1.empty() && rows.find(w1, idx1);
    // fixed format with spaces in names (:815-838): a second word that ends before column 9 and is no row name
    if ((size_t)(q - L.b) < 9 && !found1) {
      const char* ne = L.b + std::min<size_t>(10, (size_t)(L.e - L.b));
      while (ne > L.b && isWs(ne[-1])) --ne;
      P.failed = true;
      if (ne - L.b > 8) P.fail = {kReadError, "Row name \"" + std::string(L.b, ne) + "\" with spaces exceeds fixed format name length of 8"};
      else P.fail = {kReadFixedFormat, "Row name \"" + std::string(L.b, ne) + "\" with spaces: fixed format"};
      return;
    }
    if (!(w0 == cur)) {
      closeRun();
      ++runId;
      cur = w0;
      cost = 0.0;
      P.runName.push_back(w0);
      P.runBeg.push_back((int64_t)P.row.size());
    }
    const Sv w2 = nextWord(q, L.e);
    if (w2.empty()) {
      P.failed = true;
      P.fail = {kReadError, "No coefficient given for column \"" + w1.str() + "\""};
      return;
    }
    if (!found1) {
      if (P.ignoredRow++ == 0) P.firstIgnoredRow = w1;
    } else {
      entry(idx1, parseValue(w2));
    }
    const Sv w3 = nextWord(q, L.e);
    if (!w3.empty()) {
      const Sv w4 = nextWord(q, L.e);
      int32_t idx3;
      if (!rows.find(w3, idx3)) {
        if (P.ignoredRow++ == 0) P.firstIgnoredRow = w3;
      } else {
        entry(idx3, parseValue(w4));
      }
    }
  }
  closeRun();
  P.runBeg.push_back((int64_t)P.row.size());
}

// ---- RHS / RANGES / BOUNDS records (tokenised and looked up in parallel, applied in order) ---------------------
struct PairRec {  // one line of RHS or RANGES: up to two (row, value) pairs
  int32_t idx[2];   // row index, kObjRow, kFreeRow, -3 = undefined row, -4 = no second pair
  double val[2];
  const char* line; // for messages
  uint32_t len;
  uint8_t fail;     // 1 = "No bound/range given"
};
struct BoundRec {
  const char* line;  // re-tokenised in the sequential pass only when the column is not resolved here
  uint32_t len;
  int32_t col;       // >= 0: resolved (the column is a column of the COLUMNS section); -1: see `line`
  double value;
  uint8_t type;      // index into kBoundTypes, 255 = unknown
  uint8_t hasValue;
  uint8_t firstWordIsColumn;
};
// bound types of HMpsFF::parseBounds (:1346-1398): lower / upper / default value (no number on the line) /
// integral / semi-continuous or semi-integer
struct BoundType { const char* name; bool lb, ub, dflt, integral, semi; };
const BoundType kBoundTypes[11] = {
    {"UP", false, true, false, false, false}, {"LO", true, false, false, false, false}, {"FX", true, true, false, false, false},
    {"MI", true, false, true, false, false},  {"PL", false, true, true, false, false},  {"BV", true, true, true, true, false},
    {"LI", true, false, false, true, false},  {"UI", false, true, false, true, false},  {"FR", true, true, true, false, false},
    {"SI", false, true, false, true, true},   {"SC", false, true, false, false, true}};

// tokenise the data lines of a section in T pieces; the records stay in one vector per piece (file order = piece order)
template <class Rec, class Fn>
std::vector<std::vector<Rec>> parseSectionLines(const char* lo, const char* hi, const char* fileEnd, int T, Fn&& perLine) {
  std::vector<std::vector<Rec>> part((size_t)T);
  if (lo >= hi) return part;
  parallelFor(T, [&](int t) {
    const char *b, *e;
    piece(lo, hi, t, T, b, e);
    part[(size_t)t].reserve((size_t)(e - b) / 16 + 4);
    for (const char* p = b; p < e;) {
      const char* nl = lineEnd(p, fileEnd);
      const Line L = trimLine(p, nl);
      p = nl + 1;
      if (!L.skip) part[(size_t)t].push_back(perLine(L));
    }
  });
  return part;
}

struct Section {
  Key key;
  const char* lo;  // data lines: [lo, hi)
  const char* hi;
  const char* argB;  // rest of the header line (NAME / OBJSENSE arguments)
  const char* argE;
};

void addWarning(Model& m, const std::string& s) {
  ++m.numWarnings;
  m.warnings += s;
  m.warnings += '\n';
}

// gzip streams (what zstr does for the reference when it is built with zlib, HMpsFF.cpp:253-261): inflated into
// one buffer, then parsed like a mapped file.  zlib is taken from the running system with dlopen — no link-time
// dependency; without it the caller gets kReadCompressed and falls back to its own reader.
struct Zlib {
  void* h = nullptr;
  int (*inflateInit2_)(z_streamp, int, const char*, int) = nullptr;
  int (*inflate)(z_streamp, int) = nullptr;
  int (*inflateEnd)(z_streamp) = nullptr;
  int (*inflateReset)(z_streamp) = nullptr;
  Zlib() {
    for (const char* name : {"libz.so.1", "libz.so"}) {
      h = ::dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
    if (!h) return;
    inflateInit2_ = (decltype(inflateInit2_))::dlsym(h, "inflateInit2_");
    inflate = (decltype(inflate))::dlsym(h, "inflate");
    inflateEnd = (decltype(inflateEnd))::dlsym(h, "inflateEnd");
    inflateReset = (decltype(inflateReset))::dlsym(h, "inflateReset");
    if (!inflateInit2_ || !inflate || !inflateEnd || !inflateReset) { ::dlclose(h); h = nullptr; }
  }
  bool ok() const { return h != nullptr; }
};
// 0 ok, 1 corrupt stream, 2 zlib not available
int gunzip(const char* in, size_t inBytes, std::vector<char>& out) {
  static const Zlib z;
  if (!z.ok()) return 2;
  z_stream st;
  std::memset(&st, 0, sizeof(st));
  if (z.inflateInit2_(&st, 16 + MAX_WBITS, ZLIB_VERSION, (int)sizeof(z_stream)) != Z_OK) return 1;
  out.resize(std::max<size_t>(inBytes * 4, 1 << 16));
  size_t have = 0;
  st.next_in = (Bytef*)in;
  size_t left = inBytes;
  int rc = Z_OK;
  for (;;) {
    if (st.avail_in == 0 && left > 0) {
      const size_t chunk = std::min<size_t>(left, 1u << 30);
      st.avail_in = (uInt)chunk;
      left -= chunk;
    }
    if (have == out.size()) out.resize(out.size() * 2);
    const size_t room = std::min<size_t>(out.size() - have, 1u << 30);
    st.next_out = (Bytef*)out.data() + have;
    st.avail_out = (uInt)room;
    rc = z.inflate(&st, Z_NO_FLUSH);
    have += room - st.avail_out;
    if (rc == Z_STREAM_END) {
      if (st.avail_in == 0 && left == 0) break;
      if (z.inflateReset(&st) != Z_OK) { rc = Z_DATA_ERROR; break; }  // concatenated members
      continue;
    }
    if (rc != Z_OK && !(rc == Z_BUF_ERROR && st.avail_out == 0)) break;
    if (rc == Z_BUF_ERROR && st.avail_in == 0 && left == 0) break;  // truncated input
  }
  z.inflateEnd(&st);
  if (rc != Z_STREAM_END) return 1;
  out.resize(have);
  return 0;
}

// One read: the state shared by the phases below.  Errors leave through Fail (caught in run()); the lambdas that run
// on the worker threads never throw, they flag their records instead.
class Reader {
 public:
  Reader(Model& model, int threads, double timeLimit) : M(model), numThreads(threads), timeLimit_(timeLimit) {}
  ReadStatus run(const std::string& path);

 private:
  struct FlagOp { const char* pos; bool assign; bool value; };
  struct QEntry { int32_t row, col; double val; };

  ReadStatus open(const std::string& path);
  void scanSections();
  void readRows();
  void readColumns();
  void mergeColumns();
  void buildColumnState();
  void readRhsOrRanges(const Section& S);
  void readBounds(const Section& S);
  void readHessian(const Section& S);
  void finish();
  void phase(const char* what);
  int32_t findCol(Sv w, int32_t tableIdx) const;
  int32_t addCol(Sv w);

  Model& M;
  const int numThreads;
  int T = 1;
  std::chrono::steady_clock::time_point t0, lastT;
  bool timing = false;
  // the text: the mapping of the file, or the inflated gzip stream
  void* map_ = nullptr;
  size_t mapBytes_ = 0;
  std::vector<char> inflated_;
  const char* F = nullptr;
  const char* FE = nullptr;
  // HMpsFF::warning_issued_ (what turns Highs::readModel's status into kWarning) is ASSIGNED at the end of the
  // COLUMNS / RHS / BOUNDS / RANGES sections and only SET elsewhere, so an earlier warning can be forgotten; the
  // flag is replayed in file order at the end to give the same return status.
  std::vector<FlagOp> flagOps;
  std::vector<Section> sections;
  int nRowsSec = 0, nColsSec = 0;
  const char *rowsLo = nullptr, *rowsHi = nullptr, *colsLo = nullptr, *colsHi = nullptr;
  // ROWS
  std::vector<Sv> rowKeys;          // every name of the ROWS section (cost row and other N rows included)
  std::vector<int32_t> rowKeyVal;   // row index, kObjRow or kFreeRow
  std::vector<uint8_t> keyIsFree;
  std::vector<uint8_t> rowType;     // 'G','E','L' per constraint
  std::vector<Sv> rowNames;
  int32_t numRow = 0;
  NameTable rowTable;
  bool dupRowName = false;
  // COLUMNS
  std::vector<ColPiece> pieces;
  std::vector<Sv> colKeys;
  std::vector<double> colCost;
  std::vector<uint8_t> colIntegral;
  uint64_t ignoredRow = 0, dupCost = 0, dupNz = 0;
  Sv firstIgnored;
  int64_t numColFile = 0, nnz = 0;
  NameTable colTable;
  bool dupColName = false;
  std::vector<uint8_t> vtype, binary;
  std::unordered_map<std::string, int32_t> addedCols;  // columns first met in BOUNDS / Q sections (getColIdx, :491-506)
  std::vector<std::string> addedNames;
  int32_t numCol = 0;
  // RHS / RANGES / Hessian
  std::vector<uint8_t> hasRowEntry;
  bool hasObjEntry = false;
  std::vector<QEntry> qEntries;

 public:
  double timeLimit_ = 0.0;
  ~Reader() { if (map_) ::munmap(map_, mapBytes_); }
};

void Reader::phase(const char* what) {
  const auto now = std::chrono::steady_clock::now();
  // the reference checks its clock line by line (HMpsFF::timeout); the phases here take a fraction of a second each
  if (timeLimit_ > 0.0 && std::isfinite(timeLimit_) && std::chrono::duration<double>(now - t0).count() > timeLimit_)
    throw Fail{kReadTimeout, "time limit reached while reading the file"};
  if (!timing) return;
  std::fprintf(stderr, "[mps] %-28s %8.1f ms\n", what, 1e3 * std::chrono::duration<double>(now - lastT).count());
  lastT = now;
}

// tableIdx: result of the parallel lookup (-1 = not among the columns of the COLUMNS section)
int32_t Reader::findCol(Sv w, int32_t tableIdx) const {
  if (tableIdx >= 0) return tableIdx;
  if (addedCols.empty()) return -1;
  auto it = addedCols.find(w.str());
  return it == addedCols.end() ? -1 : it->second;
}
int32_t Reader::addCol(Sv w) {
  addedCols.emplace(w.str(), numCol);
  addedNames.push_back(w.str());
  M.colLower.push_back(0.0); M.colUpper.push_back(kInf); M.colCost.push_back(0.0);
  vtype.push_back(kContinuous); binary.push_back(0);
  M.aStart.push_back((int32_t)nnz);
  return numCol++;
}

ReadStatus Reader::open(const std::string& path) {
  const int fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0) { M.error = "cannot open " + path; return kReadNotFound; }
  struct stat st;
  if (::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd); M.error = "cannot stat " + path; return kReadNotFound; }
  const size_t bytes = (size_t)st.st_size;
  M.fileBytes = (int64_t)bytes;
  if (bytes == 0) { ::close(fd); M.error = "empty file"; return kReadError; }
  void* map = ::mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0);
  ::close(fd);
  if (map == MAP_FAILED) { M.error = "cannot map " + path; return kReadNotFound; }
  ::madvise(map, bytes, MADV_WILLNEED);
  map_ = map;
  mapBytes_ = bytes;
  F = (const char*)map;
  FE = F + bytes;
  if (bytes >= 2 && (unsigned char)F[0] == 0x1f && (unsigned char)F[1] == 0x8b) {
    const int zrc = gunzip(F, bytes, inflated_);
    if (zrc == 2) { M.error = "gzip stream, and zlib is not available to this reader"; return kReadCompressed; }
    if (zrc != 0) { M.error = "corrupt gzip stream"; return kReadError; }
    ::munmap(map_, mapBytes_);
    map_ = nullptr;
    if (inflated_.empty()) { M.error = "empty file"; return kReadError; }
    F = inflated_.data();
    FE = F + inflated_.size();
    M.fileBytes = (int64_t)inflated_.size();
  }
  // an explicit thread count is taken literally (the tests cut small files into many pieces with it); the
  // automatic one gives every thread at least 1 MB of text
  T = numThreads > 0 ? numThreads : (int)std::thread::hardware_concurrency();
  if (T < 1) T = 1;
  if (T > 64) T = 64;
  if (numThreads <= 0) T = (int)std::min<int64_t>(T, std::max<int64_t>(1, (int64_t)(FE - F) >> 20));
  M.threads = T;
  return kReadOk;
}

ReadStatus Reader::run(const std::string& path) {
  t0 = lastT = std::chrono::steady_clock::now();
  timing = pdlp::devEnv("PDLP_MI355X_MPS_TIMING") != nullptr;
  M = Model();
  ReadStatus status = open(path);
  if (status == kReadOk) {
    try {
      scanSections();
      readRows();
      readColumns();
      mergeColumns();
      buildColumnState();
      for (const Section& S : sections) {  // the remaining sections, in file order
        if (S.key == kRhs || S.key == kRanges) readRhsOrRanges(S);
        else if (S.key == kBounds) readBounds(S);
        else if (S.key == kQuadobj || S.key == kQmatrix || S.key == kQsection || S.key == kQcmatrix) readHessian(S);
      }
      phase("RHS/RANGES/BOUNDS/Q");
      finish();
    } catch (const Fail& f) {
      M.error = f.msg;
      status = f.status;
    }
  }
  M.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return status;
}

void Reader::scanSections() {
  // ---- pass 1: the section headers (every thread scans its piece; a header is a context-free property of a line)
  struct Header { const char* b; const char* nl; const char* after; const char* e; Key key; };
  std::vector<std::vector<Header>> hdrPart((size_t)T);
  parallelFor(T, [&](int t) {
    const char *b, *e;
    piece(F, FE, t, T, b, e);
    for (const char* p = b; p < e;) {
      const char* nl = lineEnd(p, FE);
      // cheap reject: a header's first non-blank is a letter that starts a keyword
      const char* s = p;
      while (s < nl && isWs(*s)) ++s;
      if (s < nl && *p != '*') {
        const char c = (char)(*s | 0x20);
        if (c == 'n' || c == 'o' || c == 'm' || c == 'r' || c == 'c' || c == 'b' || c == 'q' || c == 'd' || c == 'u' ||
            c == 'i' || c == 's' || c == 'g' || c == 'p' || c == 'e') {
          const Line L = trimLine(p, nl);
          if (!L.skip) {
            const char* after;
            const Key k = keyOfLine(L, after);
            if (k != kNone) hdrPart[(size_t)t].push_back({p, nl, after, L.e, k});
          }
        }
      }
      p = nl + 1;
    }
  });
  phase("map + header scan");
  std::vector<Header> hdr;
  for (auto& v : hdrPart) hdr.insert(hdr.end(), v.begin(), v.end());

  // ---- the section sequence (HMpsFF::parse, :246-354): lines outside a section are ignored, ENDATA is required
  bool ended = false;
  bool inObjsense = false;
  for (size_t i = 0; i < hdr.size() && !ended; ++i) {
    const Header& h = hdr[i];
    if (inObjsense && (h.key == kMax || h.key == kMin)) {  // parseObjsense (:562-593)
      M.sense = h.key == kMax ? -1 : 1;
      continue;
    }
    inObjsense = false;
    const char* dataLo = h.nl < FE ? h.nl + 1 : FE;
    const char* dataHi = i + 1 < hdr.size() ? hdr[i + 1].b : FE;
    switch (h.key) {
      case kEnd: ended = true; break;
      case kName: {
        const char* q = h.after;
        const Sv w = nextWord(q, h.e);
        if (!w.empty()) M.modelName = w.str();
        break;
      }
      case kObjsense: {  // Gurobi-style sense on the OBJSENSE line itself (parseDefault, :530-548)
        const char* q = h.after;
        const Sv w = nextWord(q, h.e);
        std::string u = w.str();
        for (char& c : u) c = (char)std::toupper((unsigned char)c);
        if (u == "MAX") M.sense = -1;
        else if (u == "MIN") M.sense = 1;
        inObjsense = true;
        break;
      }
      case kMax: case kMin: break;  // outside OBJSENSE: no effect (parse() falls through to parseDefault)
      case kDelayedrows: case kModelcuts: case kUsercuts: case kIndicators: case kGencons: case kPwlobj: case kPwlnam:
      case kPwlcon:
        throw Fail{kReadError, "MPS file reader cannot parse this section (DELAYEDROWS / MODELCUTS / USERCUTS / "
                                "INDICATORS / GENCONS / PWL*)"};
      default: sections.push_back({h.key, dataLo, dataHi, h.after, h.e}); break;
    }
  }
  if (!ended) throw Fail{kReadError, "no ENDATA: the MPS file is truncated"};

  auto sectionRange = [&](Key k, const char*& lo, const char*& hi, int& count) {
    count = 0;
    for (const Section& s : sections)
      if (s.key == k) {
        if (count++ == 0) { lo = s.lo; hi = s.hi; }
      }
  };
  sectionRange(kRows, rowsLo, rowsHi, nRowsSec);
  sectionRange(kCols, colsLo, colsHi, nColsSec);
  if (nRowsSec > 1 || nColsSec > 1) throw Fail{kReadError, "more than one ROWS or COLUMNS section"};
  for (const Section& s : sections)
    if (s.key == kCsection || s.key == kSets || s.key == kSos) {
      // the reference parses these and then refuses the model when they hold entries (loadProblem, :37-46)
      for (const char* p = s.lo; p < s.hi;) {
        const char* nl = lineEnd(p, FE);
        if (!trimLine(p, nl).skip) throw Fail{kReadError, "SOS and cones are not supported"};
        p = nl + 1;
      }
    }

  phase("section sequence");
}

void Reader::readRows() {
  // ---- ROWS (parseRows, :595-713) ------------------------------------------------------------------------------
  struct RowRec { char type; uint8_t bad; Sv name; Sv rest; };
  const auto rowParts = parseSectionLines<RowRec>(rowsLo, rowsHi, FE, T, [&](const Line& L) {
    RowRec r{*L.b, 0, {}, {}};
    const char* q = L.b + 1;  // the name starts right after the ONE type character (:651)
    r.name = nextWord(q, L.e);
    if (!restIsBlank(q, L.e)) { r.bad = 1; r.rest = Sv{L.b + 1, (uint32_t)(L.e - (L.b + 1))}; }
    return r;
  });
  size_t nRowRecs = 0;
  for (const auto& v : rowParts) nRowRecs += v.size();
  rowKeys.reserve(nRowRecs + 1);
  rowKeyVal.reserve(nRowRecs + 1);
  bool hasObj = false;
  Sv objName;
  for (const auto& rowRecs : rowParts)  // pieces in file order
  for (const RowRec& r : rowRecs) {
    if (r.type != 'G' && r.type != 'E' && r.type != 'L' && r.type != 'N')
      throw Fail{kReadError, "Entry \"" + std::string(1, r.type) + r.name.str() + "\" in ROWS section of MPS file is unidentified"};
    if (r.bad) {  // text after the row name: fixed format (names with spaces), :655-662
      Sv t = r.rest;
      while (t.n && isWs(*t.p)) { ++t.p; --t.n; }
      if (t.n > 8) throw Fail{kReadError, "ROWS section: name with spaces longer than 8 characters"};
      throw Fail{kReadFixedFormat, "ROWS section: names with spaces, fixed format"};
    }
    if (r.type == 'N') {
      if (!hasObj) {
        hasObj = true;
        objName = r.name;
        M.costRowLocation = numRow;
        rowKeys.push_back(r.name); rowKeyVal.push_back(kObjRow); keyIsFree.push_back(0);
      } else {
        rowKeys.push_back(r.name); rowKeyVal.push_back(kFreeRow); keyIsFree.push_back(1);
      }
      continue;
    }
    rowKeys.push_back(r.name); rowKeyVal.push_back(numRow++); keyIsFree.push_back(0);
    rowType.push_back((uint8_t)r.type);
    rowNames.push_back(r.name);
  }
  if ((int64_t)rowKeys.size() > (int64_t)0x7fffffff) throw Fail{kReadError, "too many rows"};
  Sv artificialObj{"artificial_empty_objective", 26};
  if (nRowsSec && !hasObj) {
    addWarning(M, "No objective row found");
    flagOps.push_back({rowsLo, false, true});
    rowKeys.push_back(artificialObj); rowKeyVal.push_back(kObjRow); keyIsFree.push_back(0);
  }
  M.objectiveName = hasObj ? objName.str() : "Objective";
  std::vector<uint8_t> rowDup;
  rowTable.build(rowKeys, rowKeyVal, T, &rowDup);
  dupRowName = false;
  for (size_t i = 0; i < rowDup.size() && !dupRowName; ++i) dupRowName = rowDup[i] && !keyIsFree[i];
  M.numRow = numRow;
  M.rowLower.resize((size_t)numRow);
  M.rowUpper.resize((size_t)numRow);
  for (int32_t i = 0; i < numRow; ++i) {
    const uint8_t ty = rowType[(size_t)i];
    M.rowLower[(size_t)i] = ty == 'L' ? -kInf : 0.0;
    M.rowUpper[(size_t)i] = ty == 'G' ? kInf : 0.0;
  }

  phase("ROWS + row table");
}

void Reader::readColumns() {
  // ---- COLUMNS -------------------------------------------------------------------------------------------------
  pieces = std::vector<ColPiece>((size_t)T);
  if (colsLo < colsHi)
    parallelFor(T, [&](int t) {
      const char *b, *e;
      piece(colsLo, colsHi, t, T, b, e);
      parseColumnsPiece(b, e, FE, rowTable, numRow, pieces[(size_t)t]);
    });
  for (ColPiece& P : pieces) {
    if (P.failed) throw P.fail;
    if (P.runBeg.empty()) P.runBeg.push_back(0);
  }
  phase("COLUMNS parse");
}

void Reader::mergeColumns() {
  // merge: a piece's first run continues the column of the previous piece when the names agree; MARKER lines
  // toggle the integrality of the columns CREATED after them.  Sequential over the T pieces (boundary columns,
  // marker order), parallel over the runs inside a piece.
  std::vector<int64_t> colNnz;
  std::vector<int64_t> colBase((size_t)T + 1, 0);   // columns created before piece t
  std::vector<uint8_t> integralAtStart((size_t)T, 0);
  {
    bool integral = false;
    Sv last;
    bool haveLast = false;
    for (int t = 0; t < T; ++t) {
      ColPiece& P = pieces[(size_t)t];
      ignoredRow += P.ignoredRow; dupCost += P.dupCost; dupNz += P.dupNz;
      if (firstIgnored.empty() && P.ignoredRow) firstIgnored = P.firstIgnoredRow;
      integralAtStart[(size_t)t] = integral;
      for (const ColPiece::Marker& mk : P.markers) {  // INTORG and INTEND must alternate (:793-805)
        if ((integral && mk.kind != 1) || (!integral && mk.kind != 0))
          throw Fail{kReadError, "Integrality marker error in COLUMNS section of MPS file"};
        integral = !integral;
      }
      const size_t nRun = P.runName.size();
      const bool cont = nRun > 0 && haveLast && P.runName[0] == last;
      P.firstKept = cont ? 0 : -1;
      colBase[(size_t)t + 1] = colBase[(size_t)t] + (int64_t)nRun - (cont ? 1 : 0);
      if (nRun) { last = P.runName[nRun - 1]; haveLast = true; }
    }
  }
  numColFile = colBase[(size_t)T];
  if (numColFile > 0x7ffffff0) throw Fail{kReadError, "too many columns"};
  colKeys.resize((size_t)numColFile);
  colNnz.resize((size_t)numColFile);
  colCost.resize((size_t)numColFile);
  colIntegral.resize((size_t)numColFile);
  parallelFor(T, [&](int t) {
    const ColPiece& P = pieces[(size_t)t];
    const size_t first = P.firstKept >= 0 ? 1 : 0;
    bool integral = integralAtStart[(size_t)t] != 0;
    size_t mk = 0;
    int64_t col = colBase[(size_t)t];
    for (size_t r = 0; r < P.runName.size(); ++r) {
      while (mk < P.markers.size() && P.markers[mk].runsBefore <= r) { integral = !integral; ++mk; }
      if (r < first) continue;
      colKeys[(size_t)col] = P.runName[r];
      colNnz[(size_t)col] = P.runBeg[r + 1] - P.runBeg[r];
      colCost[(size_t)col] = P.runCost[r];
      colIntegral[(size_t)col] = integral ? 1 : 0;
      ++col;
    }
  });
  {
    // boundary columns: the earlier parts' rows win; a continuation's duplicates are dropped (and counted) here
    std::vector<int32_t> gstamp;
    struct Part { int piece; size_t run; };
    std::vector<Part> parts;  // the parts of the column that is open at the end of the pieces seen so far
    for (int t = 0; t < T; ++t) {
      ColPiece& P = pieces[(size_t)t];
      const size_t nRun = P.runName.size();
      if (nRun == 0) continue;
      if (P.firstKept >= 0) {
        if (gstamp.empty()) gstamp.assign((size_t)std::max(numRow, 1), -1);
        const int64_t col = colBase[(size_t)t] - 1;
        for (const Part& pt : parts) {
          const ColPiece& Q = pieces[(size_t)pt.piece];
          const int64_t b = Q.runBeg[pt.run];
          const int64_t e = (pt.run == 0 && Q.firstKept >= 0) ? b + Q.firstKept : Q.runBeg[pt.run + 1];
          for (int64_t k = b; k < e; ++k) gstamp[(size_t)Q.row[(size_t)k]] = (int32_t)col;
        }
        int64_t w = P.runBeg[0];
        for (int64_t k = P.runBeg[0]; k < P.runBeg[1]; ++k) {
          if (gstamp[(size_t)P.row[(size_t)k]] == (int32_t)col) { ++dupNz; continue; }
          P.row[(size_t)w] = P.row[(size_t)k];
          P.val[(size_t)w] = P.val[(size_t)k];
          ++w;
        }
        // the dropped entries leave a gap inside this piece's arrays; every run keeps its own [beg, end)
        P.firstKept = w - P.runBeg[0];
        colNnz[(size_t)col] += P.firstKept;
        if (P.runCost[0] != 0.0 || P.runCost[0] != P.runCost[0]) {
          if (colCost[(size_t)col] != 0.0 || colCost[(size_t)col] != colCost[(size_t)col]) ++dupCost;
          else colCost[(size_t)col] = P.runCost[0];
        }
        if (nRun == 1) { parts.push_back({t, 0}); continue; }  // the column is still open
      }
      parts.clear();
      parts.push_back({t, nRun - 1});
    }
  }
  phase("COLUMNS merge");
  nnz = 0;
  M.aStart.resize((size_t)numColFile + 1);
  for (int64_t j = 0; j < numColFile; ++j) {
    M.aStart[(size_t)j] = (int32_t)std::min<int64_t>(nnz, 0x7fffffff);
    nnz += colNnz[(size_t)j];
  }
  if (nnz > 0x7fffffff) throw Fail{kReadError, "more than 2^31 - 1 nonzeros"};
  M.aStart[(size_t)numColFile] = (int32_t)nnz;
  M.aIndex.resize((size_t)nnz);
  M.aValue.resize((size_t)nnz);
  parallelFor(T, [&](int t) {
    const ColPiece& P = pieces[(size_t)t];
    // a continuation is appended behind what the earlier parts of its column hold: start of the column + its
    // length so far = start of the next column - what this piece and the later ones contribute
    int64_t col = colBase[(size_t)t] - 1;
    for (size_t r = 0; r < P.runName.size(); ++r) {
      const bool cont = r == 0 && P.firstKept >= 0;
      const int64_t b = P.runBeg[r];
      const int64_t n = cont ? P.firstKept : P.runBeg[r + 1] - b;
      int64_t dst;
      if (cont) {
        int64_t later = 0;  // entries of the same column in later pieces (a column spanning more than two pieces)
        for (int u = t + 1; u < T && P.runName.size() == 1; ++u) {  // (only if this piece lies inside the column)
          const ColPiece& Q = pieces[(size_t)u];
          if (Q.runName.empty()) continue;
          if (Q.firstKept < 0) break;
          later += Q.firstKept;
          if (Q.runName.size() > 1) break;
        }
        dst = (int64_t)M.aStart[(size_t)col + 1] - later - n;
      } else {
        dst = M.aStart[(size_t)++col];
      }
      if (n > 0) {
        std::memcpy(&M.aIndex[(size_t)dst], &P.row[(size_t)b], sizeof(int32_t) * (size_t)n);
        std::memcpy(&M.aValue[(size_t)dst], &P.val[(size_t)b], sizeof(double) * (size_t)n);
      }
    }
  });
  phase("COLUMNS copy");
}

void Reader::buildColumnState() {
  if (nColsSec) flagOps.push_back({colsLo, true, ignoredRow || dupCost || dupNz});
  if (ignoredRow || dupCost || dupNz)
    addWarning(M, "COLUMNS section: ignored " + std::to_string(ignoredRow) + " undefined rows " + std::to_string(dupCost) +
                      " duplicate cost values and " + std::to_string(dupNz) + " duplicate matrix values" +
                      (ignoredRow ? " (first undefined row \"" + firstIgnored.str() + "\")" : std::string()));

  // column name table; later sections may add columns (getColIdx(name, add_if_new), :491-506)
  std::vector<int32_t> colKeyVal((size_t)numColFile);
  for (int64_t j = 0; j < numColFile; ++j) colKeyVal[(size_t)j] = (int32_t)j;
  std::vector<uint8_t> colDup;
  colTable.build(colKeys, colKeyVal, T, &colDup);
  dupColName = false;
  for (uint8_t d : colDup) if (d) { dupColName = true; break; }
  M.colLower.assign((size_t)numColFile, 0.0);
  M.colUpper.assign((size_t)numColFile, kInf);
  M.colCost = std::move(colCost);
  vtype.assign((size_t)numColFile, kContinuous);
  binary.assign((size_t)numColFile, 0);  // integer columns of the COLUMNS section are binary until a bound says otherwise (:881)
  for (int64_t j = 0; j < numColFile; ++j) { vtype[(size_t)j] = colIntegral[(size_t)j] ? kInteger : kContinuous; binary[(size_t)j] = colIntegral[(size_t)j]; }
  numCol = (int32_t)numColFile;
  phase("column table");
}

void Reader::readRhsOrRanges(const Section& S) {
  const bool isRhs = S.key == kRhs;
  const std::string& mpsName = M.modelName;
  const auto parts = parseSectionLines<PairRec>(S.lo, S.hi, FE, T, [&](const Line& L) {
    PairRec r;
    r.idx[0] = r.idx[1] = -4;
    r.val[0] = r.val[1] = 0.0;
    r.fail = 0;
    r.line = L.b;
    r.len = (uint32_t)(L.e - L.b);
    const char* q = L.b;
    Sv w0 = nextWord(q, L.e);
    int32_t v;
    Sv marker;
    // RHS only: the set name may be missing (SIF), recognised by the first word being a row name (:1120-1125)
    if (isRhs && rowTable.find(w0, v)) marker = w0;
    else marker = nextWord(q, L.e);
    Sv word = nextWord(q, L.e);
    if (word.empty()) { r.fail = 1; return r; }
    bool found = rowTable.find(marker, v);
    if (!found && isRhs && !mpsName.empty() && marker.n == mpsName.size() && std::memcmp(marker.p, mpsName.data(), marker.n) == 0) {
      marker = word;  // SIF: the model name in front of the entry (:1145-1162)
      word = nextWord(q, L.e);
      if (word.empty()) { r.fail = 1; return r; }
      found = rowTable.find(marker, v);
    }
    r.idx[0] = found ? v : -3;
    r.val[0] = parseValue(word);
    const Sv m2 = nextWord(q, L.e);
    if (!m2.empty()) {
      const Sv w2 = nextWord(q, L.e);
      if (!isRhs && w2.empty()) { r.fail = 1; return r; }  // RANGES checks the second value (:1669-1675)
      r.idx[1] = rowTable.find(m2, v) ? v : -3;
      r.val[1] = parseValue(w2);
    }
    return r;
  });
  hasRowEntry.assign((size_t)std::max(numRow, 1), 0);
  if (isRhs) hasObjEntry = false;
  uint64_t ignored = 0, dup = 0;
  for (const auto& recs : parts)  // pieces in file order
  for (const PairRec& r : recs) {
    if (r.fail) throw Fail{kReadError, std::string(isRhs ? "No bound given in RHS line \"" : "No range given in RANGES line \"") + std::string(r.line, r.len) + "\""};
    for (int k = 0; k < 2; ++k) {
      const int32_t idx = r.idx[k];
      if (idx == -4) continue;
      if (idx == -3) { ++ignored; continue; }
      const double val = r.val[k];
      if (isRhs) {
        if (idx >= 0) {
          if (hasRowEntry[(size_t)idx]) { ++dup; continue; }
          const uint8_t ty = rowType[(size_t)idx];
          if (ty == 'E' || ty == 'L') M.rowUpper[(size_t)idx] = val;
          if (ty == 'E' || ty == 'G') M.rowLower[(size_t)idx] = val;
          hasRowEntry[(size_t)idx] = 1;
        } else {  // the cost row: objective offset (:1078-1083); other N rows take the same branch in the reference
          if (hasObjEntry) { ++dup; continue; }
          M.offset = -val;
          hasObjEntry = true;
        }
      } else {
        if (idx < 0) { ++ignored; continue; }
        if (hasRowEntry[(size_t)idx]) { ++dup; continue; }
        const uint8_t ty = rowType[(size_t)idx];
        if ((ty == 'E' && val < 0) || ty == 'L') M.rowLower[(size_t)idx] = M.rowUpper[(size_t)idx] - std::fabs(val);
        else if ((ty == 'E' && val > 0) || ty == 'G') M.rowUpper[(size_t)idx] = M.rowLower[(size_t)idx] + std::fabs(val);
        hasRowEntry[(size_t)idx] = 1;
      }
    }
  }
  flagOps.push_back({S.lo, true, ignored || dup});
  if (ignored || dup)
    addWarning(M, std::string(isRhs ? "RHS" : "RANGES") + " section: ignored " + std::to_string(ignored) +
                      " undefined rows and " + std::to_string(dup) + " duplicate values");
}

void Reader::readBounds(const Section& S) {
  const auto parts = parseSectionLines<BoundRec>(S.lo, S.hi, FE, T, [&](const Line& L) {
    BoundRec r;
    r.line = L.b;
    r.len = (uint32_t)(L.e - L.b);
    r.col = -1;
    r.value = 0.0;
    r.hasValue = 0;
    r.firstWordIsColumn = 0;
    const char* q = L.b;
    const Sv ty = nextWord(q, L.e);
    r.type = 255;
    if (ty.n == 2)
      for (int k = 0; k < 11; ++k)
        if (ty.p[0] == kBoundTypes[k].name[0] && ty.p[1] == kBoundTypes[k].name[1]) r.type = (uint8_t)k;
    if (r.type == 255) return r;
    const Sv w1 = nextWord(q, L.e);
    const Sv w2 = nextWord(q, L.e);
    int32_t v;
    // the bound-set name may be missing (SIF): then the first word is a column (:1405-1415)
    if (!w1.empty() && colTable.find(w1, v)) {
      r.col = v;
      r.firstWordIsColumn = 1;
      r.hasValue = !w2.empty();
      if (!kBoundTypes[r.type].dflt) r.value = parseValue(w2);
    } else if (!w2.empty() && colTable.find(w2, v)) {
      r.col = v;  // provisional: holds unless an EARLIER bounds line created a column called w1 (checked below)
      const Sv w3 = nextWord(q, L.e);
      r.hasValue = !w3.empty();
      if (!kBoundTypes[r.type].dflt) r.value = parseValue(w3);
    }
    return r;
  });
  std::vector<uint8_t> hasLower((size_t)numCol, 0), hasUpper((size_t)numCol, 0);
  uint64_t dup = 0, fractional = 0;
  for (const auto& recs : parts)  // pieces in file order
  for (const BoundRec& r : recs) {
    if (r.type == 255) {
      const char* q = r.line;
      throw Fail{kReadError, "Entry in BOUNDS section of MPS file is of type \"" + nextWord(q, r.line + r.len).str() + "\""};
    }
    const BoundType& bt = kBoundTypes[r.type];
    int32_t col = r.col;
    double value = r.value;
    bool hasValue = r.hasValue != 0;
    Sv marker;
    if (col < 0 || (!r.firstWordIsColumn && !addedCols.empty())) {
      // rare: a column this section introduces, or the names of such columns have to be consulted first
      const char* q = r.line;
      const char* e = r.line + r.len;
      nextWord(q, e);
      const Sv w1 = nextWord(q, e);
      const Sv w2 = nextWord(q, e);
      const Sv w3 = nextWord(q, e);
      int32_t v;
      col = findCol(w1, colTable.find(w1, v) ? v : -1);
      marker = w1;
      Sv valueWord = w2;
      if (col < 0) {
        marker = w2;
        valueWord = w3;
        col = findCol(w2, !w2.empty() && colTable.find(w2, v) ? v : -1);
        if (col < 0) {
          col = addCol(marker);
          hasLower.push_back(0);
          hasUpper.push_back(0);
        }
      }
      hasValue = !valueWord.empty();
      value = parseValue(valueWord);
    }
    if ((bt.lb && hasLower[(size_t)col]) || (bt.ub && hasUpper[(size_t)col])) { ++dup; continue; }
    if (bt.dflt) {
      if (bt.integral) {  // BV
        vtype[(size_t)col] = kInteger;
        binary[(size_t)col] = 1;
        M.colUpper[(size_t)col] = 1.0;
      } else {
        binary[(size_t)col] = 0;
        if (bt.lb) M.colLower[(size_t)col] = -kInf;
        if (bt.ub) M.colUpper[(size_t)col] = kInf;
      }
      if (bt.lb) hasLower[(size_t)col] = 1;
      if (bt.ub) hasUpper[(size_t)col] = 1;
      continue;
    }
    if (!hasValue) throw Fail{kReadError, std::string("No bound given in BOUNDS line \"") + std::string(r.line, r.len) + "\""};
    if (bt.integral) {
      if (value - (double)(int32_t)value != 0.0) ++fractional;
      vtype[(size_t)col] = bt.semi ? kSemiInteger : kInteger;
    } else if (bt.semi) {
      vtype[(size_t)col] = kSemiContinuous;
    }
    if (bt.lb) { M.colLower[(size_t)col] = value; hasLower[(size_t)col] = 1; }
    if (bt.ub) { M.colUpper[(size_t)col] = value; hasUpper[(size_t)col] = 1; }
    binary[(size_t)col] = 0;
  }
  flagOps.push_back({S.lo, true, dup || fractional});
  if (dup || fractional)
    addWarning(M, "BOUNDS section: ignored " + std::to_string(dup) + " duplicate values and " + std::to_string(fractional) +
                      " fractional integer bounds");
}

void Reader::readHessian(const Section& S) {
  if (S.key == kQsection || S.key == kQcmatrix) {
    // parseQuadRows (:1745-1801): the section names a row; the cost row's section is the objective Hessian, an
    // undefined or free row's section is skipped, a constraint's one makes the model a QCP (refused, :32-36)
    const char* q = S.argB;
    const Sv rn = nextWord(q, S.argE);
    if (rn.empty()) throw Fail{kReadError, "No row name given in argument of QSECTION / QCMATRIX"};
    int32_t ri;
    if (!rowTable.find(rn, ri)) {
      addWarning(M, "Row name \"" + rn.str() + "\" in QSECTION / QCMATRIX section is not defined: ignored");
      flagOps.push_back({S.lo, false, true});
      return;
    }
    if (ri == kFreeRow) return;
    if (ri >= 0) throw Fail{kReadError, "Quadratic rows not supported by HiGHS"};
  }
  // parseQuadMatrix (:1803-1889): QUADOBJ / QSECTION list one triangle, every off-diagonal entry also defines its mirror
  const bool triangular = S.key == kQuadobj || S.key == kQsection;
  for (const char* p = S.lo; p < S.hi;) {
    const char* nl = lineEnd(p, FE);
    const Line L = trimLine(p, nl);
    p = nl + 1;
    if (L.skip) continue;
    const char* q = L.b;
    const Sv cn = nextWord(q, L.e);
    int32_t v;
    int32_t col = findCol(cn, colTable.find(cn, v) ? v : -1);
    if (col < 0) col = addCol(cn);
    for (int k = 0; k < 2; ++k) {
      const Sv rn = nextWord(q, L.e);
      if (rn.empty()) break;
      const Sv cv = nextWord(q, L.e);
      if (cv.empty()) throw Fail{kReadError, "Hessian section has no coefficient for entry \"" + rn.str() + "\" in column \"" + cn.str() + "\""};
      int32_t row = findCol(rn, colTable.find(rn, v) ? v : -1);
      if (row < 0) row = addCol(rn);
      const double c = parseValue(cv);
      if (c != 0.0 || c != c) {
        qEntries.push_back({row, col, c});
        if (triangular && row != col) qEntries.push_back({col, row, c});
      }
    }
  }
}

void Reader::finish() {
  // columns that are still binary by default (parse(), :326-332)
  for (int32_t j = 0; j < numCol; ++j)
    if (binary[(size_t)j]) { M.colLower[(size_t)j] = 0.0; M.colUpper[(size_t)j] = 1.0; }
  M.numCol = numCol;
  bool isMip = false;
  for (uint8_t t : vtype) if (t != kContinuous) { isMip = true; break; }
  if (isMip) M.integrality = std::move(vtype);

  // Hessian: square, column-wise, entry order inside a column (fillHessian)
  if (!qEntries.empty()) {
    M.qDim = numCol;
    M.qStart.assign((size_t)numCol + 1, 0);
    for (const QEntry& q : qEntries) ++M.qStart[(size_t)q.col + 1];
    for (int32_t j = 0; j < numCol; ++j) M.qStart[(size_t)j + 1] += M.qStart[(size_t)j];
    std::vector<int32_t> fill(M.qStart.begin(), M.qStart.end() - 1);
    M.qIndex.resize(qEntries.size());
    M.qValue.resize(qEntries.size());
    for (const QEntry& q : qEntries) {
      const int32_t k = fill[(size_t)q.col]++;
      M.qIndex[(size_t)k] = q.row;
      M.qValue[(size_t)k] = q.val;
    }
  }

  // names (cleared when duplicated, loadProblem :63-80)
  if (dupRowName) addWarning(M, "Linear constraints have duplicate names: row names dropped");
  else {
    M.rowNameStart.resize(rowNames.size() + 1);
    size_t o = 0;
    for (size_t i = 0; i < rowNames.size(); ++i) { M.rowNameStart[i] = (int64_t)o; o += rowNames[i].n + 1; }
    M.rowNameStart[rowNames.size()] = (int64_t)o;
    M.rowNamePool.resize(o);
    parallelFor(T, [&](int t) {
      const size_t n = rowNames.size();
      for (size_t i = n * t / T; i < n * (t + 1) / T; ++i) {
        char* d = &M.rowNamePool[(size_t)M.rowNameStart[i]];
        std::memcpy(d, rowNames[i].p, rowNames[i].n);
        d[rowNames[i].n] = '\0';
      }
    });
  }
  if (dupColName) addWarning(M, "Variables have duplicate names: column names dropped");
  else {
    M.colNameStart.resize((size_t)numCol + 1);
    size_t o = 0;
    for (size_t j = 0; j < colKeys.size(); ++j) { M.colNameStart[j] = (int64_t)o; o += colKeys[j].n + 1; }
    for (size_t j = 0; j < addedNames.size(); ++j) { M.colNameStart[colKeys.size() + j] = (int64_t)o; o += addedNames[j].size() + 1; }
    M.colNameStart[(size_t)numCol] = (int64_t)o;
    M.colNamePool.resize(o);
    parallelFor(T, [&](int t) {
      const size_t n = colKeys.size();
      for (size_t j = n * t / T; j < n * (t + 1) / T; ++j) {
        char* d = &M.colNamePool[(size_t)M.colNameStart[j]];
        std::memcpy(d, colKeys[j].p, colKeys[j].n);
        d[colKeys[j].n] = '\0';
      }
    });
    for (size_t j = 0; j < addedNames.size(); ++j)
      std::memcpy(&M.colNamePool[(size_t)M.colNameStart[colKeys.size() + j]], addedNames[j].c_str(), addedNames[j].size() + 1);
  }
  phase("finalise + names");
  std::stable_sort(flagOps.begin(), flagOps.end(), [](const FlagOp& x, const FlagOp& y) { return x.pos < y.pos; });
  for (const FlagOp& op : flagOps) M.warningIssued = op.assign ? op.value : (M.warningIssued || op.value);
  if (dupRowName || dupColName) M.warningIssued = true;
}

}  // namespace

// ==================================================================================================================
ReadStatus readMps(const std::string& path, int numThreads, Model& M, double timeLimit) {
  Reader reader(M, numThreads, timeLimit);
  return reader.run(path);
}

void lowerTriangle(const Model& m, std::vector<int32_t>& start, std::vector<int32_t>& index, std::vector<double>& value) {
  const int32_t dim = m.qDim;
  start.assign((size_t)dim + 1, 0);
  index.clear();
  value.clear();
  if (dim == 0) return;
  // (row >= col) entries of (Q + Q')/2: an entry (i, j) of the square matrix contributes half to the lower-triangle
  // position (max, min); diagonal entries count in full.  Duplicates are summed.
  struct E { int32_t col, row; double v; };
  std::vector<E> ent;
  ent.reserve(m.qIndex.size());
  for (int32_t j = 0; j < dim; ++j)
    for (int32_t k = m.qStart[(size_t)j]; k < m.qStart[(size_t)j + 1]; ++k) {
      const int32_t i = m.qIndex[(size_t)k];
      const double v = m.qValue[(size_t)k];
      if (i == j) ent.push_back({j, i, v});
      else ent.push_back({std::min(i, j), std::max(i, j), 0.5 * v});
    }
  std::stable_sort(ent.begin(), ent.end(), [](const E& a, const E& b) { return a.col != b.col ? a.col < b.col : a.row < b.row; });
  for (size_t k = 0; k < ent.size();) {
    size_t e = k;
    double s = 0.0;
    while (e < ent.size() && ent[e].col == ent[k].col && ent[e].row == ent[k].row) s += ent[e++].v;
    index.push_back(ent[k].row);
    value.push_back(s);
    ++start[(size_t)ent[k].col + 1];
    k = e;
  }
  for (int32_t j = 0; j < dim; ++j) start[(size_t)j + 1] += start[(size_t)j];
}

}  // namespace mps
}  // namespace pdlp
