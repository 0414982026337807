// dlsa_amd/csrc/sw_db.cpp -- the resident target database of libsw_hip.so: 2-bit packing into length-sorted
// groups, the ticket-ordered unit table, the inter-task score pass, re-run rules, on-device top-k selection and
// the search entry points of include/sw_hip.h.  Host code over the kernels of sw_score.hip / sw_topk.hip /
// sw_exact.hip; nothing here computes a cell on the CPU.
//
// What it replaces: for a work package, the per-pair loop of internal/worker/worker_impl.go:108-168 and the
// client-side top-k of cli/__main__.py:153-173.
#include <algorithm>
#include <array>
#include <atomic>
#include <cerrno>
#include <cstring>
#include <memory>
#include <numeric>
#include <thread>

#include "sw_host.h"

using namespace swhip;

struct sw_db {
  Context* ctx = nullptr;          // the device this database is resident on
  uint64_t n = 0, residues = 0;
  std::vector<uint64_t> off;       // n + 1
  std::string flat;                // original bytes
  std::vector<uint8_t> packed_ok;  // per target: only A, C, G, T
  // packed part
  std::vector<uint32_t> slot_target;  // per slot: database index or UINT32_MAX
  std::vector<ScoreGroup> groups;
  uint32_t* d_words = nullptr;
  uint32_t* d_lens = nullptr;
  uint32_t* d_slot_target = nullptr;  // per slot: database index (UINT32_MAX: padding)
  uint64_t packed_residues = 0;       // residues of the packed targets
  ScoreGroup* d_groups = nullptr;
  uint32_t* d_elem_group = nullptr;  // per strip of the stream: its group
  uint32_t n_elems = 0;              // strips in the stream
  uint32_t n_half_elems = 0;         // ... of which run the half-width row loop (ScoreGroup::last_cols <= 8)
  std::vector<uint32_t> len64;       // sorted distinct lengths of packed targets that are multiples of 64 (query batching)
  // ticket-ordered unit table, rebuilt when the lane count / ring geometry of a search differs (build_units)
  ScoreUnit* d_units = nullptr;
  uint32_t n_units = 0, units_S = 0, units_R = 0, units_W = 0;
  uint64_t units_chained = 0;        // units that hand their last H column to the next unit through HBM
  std::shared_ptr<const std::vector<ScoreUnit>> h_units;  // host copy of the table currently on the device (a pass
                                                          // in flight keeps its own reference: PersistPass)
  std::vector<uint32_t> generic;  // targets scored by the exact kernel only
  uint32_t max_len = 0;
  // a TEMPORARY database (the work-package fast path builds one per target set and drops it a few milliseconds
  // later): its device arrays are the context's (Context::tdb), not allocations of its own -- hipMalloc and above all
  // hipFree, which waits for every stream of the device, would serialise the sets that run side by side
  bool borrowed = false;
  // counters of THIS database's last search (phase A resets, phase B adds): several logical shards may share one
  // device context, whose own counters then belong to whichever shard ran last
  sw_kernel_stats last_stats{};
};

namespace {

inline int nt_code(unsigned char ch) {
  switch (ch) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return -1;
  }
}

void db_release(sw_db* db) {
  if (!db) return;
  if (db->borrowed) {  // (caller holds the context's lock)
    if (db->ctx) db->ctx->tdb.busy = false;
    delete db;
    return;
  }
  if (db->d_words) (void)hipFree(db->d_words);
  if (db->d_lens) (void)hipFree(db->d_lens);
  if (db->d_slot_target) (void)hipFree(db->d_slot_target);
  if (db->d_groups) (void)hipFree(db->d_groups);
  if (db->d_elem_group) (void)hipFree(db->d_elem_group);
  if (db->d_units) (void)hipFree(db->d_units);
  delete db;
}

struct DbDeleter {
  void operator()(sw_db* d) const { db_release(d); }
};
// strips that run the half-width row loop: the last strip of a group whose targets all end within its first 8 columns
// ... and the lengths that are multiples of 64 (per-slot lengths `lens`, 0 = padding)
void count_half_strips(sw_db* db, const std::vector<uint32_t>& lens) {
  db->n_half_elems = 0;
  for (const ScoreGroup& G : db->groups) db->n_half_elems += (G.last_cols >= 1 && G.last_cols <= kScoreR / 2);
  db->len64.clear();
  for (uint32_t len : lens)
    if (len && len % 64 == 0) db->len64.push_back(len);
  std::sort(db->len64.begin(), db->len64.end());
  db->len64.erase(std::unique(db->len64.begin(), db->len64.end()), db->len64.end());
}

// The packed part of a database goes to the device (caller holds the context's lock, its device is current).
bool db_upload(sw_db* db, const std::vector<uint32_t>& words, const std::vector<uint32_t>& lens,
               const std::vector<uint32_t>& elem_group) {
  const size_t ngroups = db->groups.size();
  if (!ngroups) return true;
  if (db->borrowed) {
    Context& c = *db->ctx;
    struct Part {
      DevBuf* buf;
      const void* src;
      size_t bytes;
      void** dst;
    } parts[5] = {{&c.tdb.words, words.data(), words.size() * 4, (void**)&db->d_words},
                  {&c.tdb.lens, lens.data(), lens.size() * 4, (void**)&db->d_lens},
                  {&c.tdb.slot_target, db->slot_target.data(), lens.size() * 4, (void**)&db->d_slot_target},
                  {&c.tdb.groups, db->groups.data(), ngroups * sizeof(ScoreGroup), (void**)&db->d_groups},
                  {&c.tdb.elem_group, elem_group.data(), elem_group.size() * 4, (void**)&db->d_elem_group}};
    for (Part& p : parts) {
      if (p.buf->ensure(std::max<size_t>(p.bytes, 16))) return false;
      *p.dst = p.buf->p;
      if (hipMemcpyAsync(p.buf->p, p.src, p.bytes, hipMemcpyHostToDevice, c.stream) != hipSuccess) {
        (void)hipStreamSynchronize(c.stream);
        fail(SW_E_HIP, "hipMemcpy failed for the database");
        return false;
      }
    }
    if (hipStreamSynchronize(c.stream) != hipSuccess) {  // (the sources are the caller's temporaries)
      fail(SW_E_HIP, "hipMemcpy failed for the database");
      return false;
    }
    return true;
  }
  if (hipMalloc(&db->d_words, words.size() * 4) != hipSuccess || hipMalloc(&db->d_lens, lens.size() * 4) != hipSuccess ||
      hipMalloc(&db->d_slot_target, lens.size() * 4) != hipSuccess ||
      hipMalloc(&db->d_groups, ngroups * sizeof(ScoreGroup)) != hipSuccess ||
      hipMalloc(&db->d_elem_group, elem_group.size() * 4) != hipSuccess) {
    fail(SW_E_NOMEM, "hipMalloc failed for the database");
    return false;
  }
  if (hipMemcpy(db->d_words, words.data(), words.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(db->d_lens, lens.data(), lens.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(db->d_slot_target, db->slot_target.data(), lens.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(db->d_groups, db->groups.data(), ngroups * sizeof(ScoreGroup), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(db->d_elem_group, elem_group.data(), elem_group.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
    fail(SW_E_HIP, "hipMemcpy failed for the database");
    return false;
  }
  return true;
}
// 16 residues out of {A, C, G, T} -> 2 bits each, residue r in bits [2r, 2r + 2).  ((c >> 1) ^ (c >> 2)) & 3 maps
// 'A' 'C' 'G' 'T' to 0 1 2 3; eight bytes at a time, the 2-bit fields are then gathered by three shift-or steps.
inline uint32_t pack8(const unsigned char* p) {
  uint64_t x;
  memcpy(&x, p, 8);  // (little-endian: byte r is residue r)
  x = ((x >> 1) ^ (x >> 2)) & 0x0303030303030303ull;
  x = (x | (x >> 6)) & 0x000f000f000f000full;
  x = (x | (x >> 12)) & 0x000000ff000000ffull;
  x = (x | (x >> 24)) & 0xffffull;
  return (uint32_t)x;
}
inline uint32_t pack16(const unsigned char* p) { return pack8(p) | (pack8(p + 8) << 16); }

// Caller holds the context's lock (its device is current).
// known_acgt: the caller has checked that every sequence consists of A, C, G, T only (the work-package fast path).
sw_db* db_build(Context* ctx, std::string&& flat, std::vector<uint64_t>&& off, bool known_acgt = false, bool temporary = false) {
  std::unique_ptr<sw_db, DbDeleter> holder(new sw_db);  // an exception below must not leak the device arrays
  sw_db* db = holder.get();
  db->ctx = ctx;
  if (temporary && !ctx->tdb.busy) {  // (one temporary database per context at a time; a second one allocates)
    ctx->tdb.busy = true;
    db->borrowed = true;
  }
  db->n = off.size() - 1;
  db->flat = std::move(flat);
  db->off = std::move(off);
  db->residues = db->off.back() - db->off.front();
  StageTimer tm(db->n);
  db->packed_ok.assign(db->n, 1);
  for (uint64_t i = 0; i < db->n; i++) {
    const uint64_t len = db->off[i + 1] - db->off[i];
    if (len >= 0x7fffff00ull) {
      fail(SW_E_ARG, "sequence too long");
      return nullptr;
    }
    db->max_len = std::max<uint32_t>(db->max_len, (uint32_t)len);
  }
  // per target: 1 = only A/C/G/T (packable), 0 = something else (exact kernel), 2 = NUL inside
  static const std::array<int8_t, 256> kCode = [] {
    std::array<int8_t, 256> t;
    t.fill(-1);
    t['A'] = 0, t['C'] = 1, t['G'] = 2, t['T'] = 3;
    return t;
  }();
  std::atomic<bool> has_nul{false};
  parallel_ranges(known_acgt ? 0 : db->n, db->n >= 4096 ? db->flat.size() + ((size_t)1 << 20) : 0, [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; i++) {
      const unsigned char* s = (const unsigned char*)db->flat.data() + db->off[i];
      const uint64_t len = db->off[i + 1] - db->off[i];
      // (comparisons instead of a table look-up: the compiler turns this into 32 bytes per instruction)
      unsigned bad = 0, nul = 0;
      for (uint64_t k = 0; k < len; k++) {
        const unsigned char ch = s[k];
        bad |= !((ch == 'A') | (ch == 'C') | (ch == 'G') | (ch == 'T'));
        nul |= (ch == 0);
      }
      if (nul) has_nul = true;
      db->packed_ok[i] = !bad;
    }
  });
  if (has_nul) {
    fail(SW_E_ARG, "NUL byte inside a sequence");
    return nullptr;
  }
  tm.lap("    db: scan");
  // sequences are text like the per-pair inputs (bindings.rs:125-129): anything non-ASCII must be valid UTF-8
  for (uint64_t i = 0; i < db->n; i++) {
    if (db->packed_ok[i]) continue;
    Decoded d;
    if (!decode(db->flat.data() + db->off[i], (size_t)(db->off[i + 1] - db->off[i]), d)) {
      fail(SW_E_ARG, "sequence " + std::to_string(i) + " is not valid UTF-8");
      return nullptr;
    }
  }
  std::vector<uint32_t> packable;
  for (uint64_t i = 0; i < db->n; i++) {
    if (db->packed_ok[i] && db->off[i + 1] > db->off[i]) packable.push_back((uint32_t)i);
    else db->generic.push_back((uint32_t)i);
  }
  // length-sorted (descending, stable on index) groups of 128
  if ((uint64_t)db->max_len <= 8 * (uint64_t)packable.size() + 65536) {
    // counting sort by length (10^7 reads: 0.4 s of comparison sort otherwise): positions from the longest down,
    // filled in index order, which keeps equal lengths stable
    std::vector<uint64_t> at((size_t)db->max_len + 2, 0);
    for (uint32_t i : packable) at[(size_t)(db->off[i + 1] - db->off[i])]++;
    uint64_t pos = 0;
    for (size_t len = at.size(); len-- > 0;) {
      const uint64_t c = at[len];
      at[len] = pos;
      pos += c;
    }
    std::vector<uint32_t> sorted(packable.size());
    for (uint32_t i : packable) sorted[at[(size_t)(db->off[i + 1] - db->off[i])]++] = i;
    packable.swap(sorted);
  } else {
    std::stable_sort(packable.begin(), packable.end(), [&](uint32_t a, uint32_t b) {
      return (db->off[a + 1] - db->off[a]) > (db->off[b + 1] - db->off[b]);
    });
  }
  const size_t ngroups = (packable.size() + kGroupTargets - 1) / kGroupTargets;
  db->groups.resize(ngroups);
  db->slot_target.assign(ngroups * kGroupTargets, UINT32_MAX);
  std::vector<uint32_t> lens(ngroups * kGroupTargets, 0);
  size_t total_strips = 0;
  for (size_t g = 0; g < ngroups; g++) {
    uint32_t maxlen = 0;
    for (int s = 0; s < kGroupTargets; s++) {
      size_t k = g * kGroupTargets + s;
      if (k >= packable.size()) break;
      uint32_t ti = packable[k];
      uint32_t len = (uint32_t)(db->off[ti + 1] - db->off[ti]);
      db->slot_target[k] = ti;
      lens[k] = len;
      db->packed_residues += len;
      maxlen = std::max(maxlen, len);
    }
    if (total_strips > 0xfff00000ull) {
      fail(SW_E_ARG, "database too large for one device (more than 2^32 strips)");
      return nullptr;
    }
    db->groups[g].strip_base = (uint32_t)total_strips;
    db->groups[g].nstrips = (maxlen + kScoreR - 1) / kScoreR;
    db->groups[g].first_slot = (uint32_t)(g * kGroupTargets);
    db->groups[g].last_cols = maxlen ? maxlen - kScoreR * (db->groups[g].nstrips - 1) : 0;
    total_strips += db->groups[g].nstrips;
  }
  db->n_elems = (uint32_t)total_strips;
  count_half_strips(db, lens);
  tm.lap("    db: sort + groups");
  // the strip stream: strip e = dwords [e*128, e*128+128), [half][lane], 16 residues (2 bits each) per dword
  std::vector<uint32_t> words(std::max<size_t>(total_strips * kGroupTargets, 1), 0);
  std::vector<uint32_t> elem_group(std::max<size_t>(total_strips, 1), 0);
  parallel_ranges(ngroups, ngroups >= 32 ? db->flat.size() + ((size_t)1 << 20) : 0, [&](size_t glo, size_t ghi) {
    for (size_t g = glo; g < ghi; g++) {
      const ScoreGroup& G = db->groups[g];
      for (uint32_t st = 0; st < G.nstrips; st++) elem_group[G.strip_base + st] = (uint32_t)g;
      // in blocks of 64 strips (1024 columns): the words of a strip lie 512 bytes apart, so one target's words of a
      // block land in a 32 KB region that stays in the cache while the other 127 targets fill it in
      constexpr uint32_t kStripBlock = 64;
      for (uint32_t sb = 0; sb < G.nstrips; sb += kStripBlock) {
        const uint32_t c0 = sb * kScoreR, c1 = std::min<uint32_t>(G.nstrips, sb + kStripBlock) * kScoreR;
        for (int s = 0; s < kGroupTargets; s++) {
          size_t k = g * kGroupTargets + s;
          uint32_t ti = db->slot_target[k];
          if (ti == UINT32_MAX) continue;
          const unsigned char* seq = (const unsigned char*)db->flat.data() + db->off[ti];
          const uint32_t len = std::min(lens[k], c1);
          if (len <= c0) continue;
          uint32_t* dst = words.data() + ((size_t)G.strip_base + sb) * kGroupTargets + (size_t)(s / 64) * 64 + (s % 64);
          uint32_t col = c0;
          for (; col + kScoreR <= len; col += kScoreR, dst += kGroupTargets) *dst = pack16(seq + col);
          if (col < len) {
            uint32_t w = 0;
            for (int r = 0; col + r < len; r++) w |= (uint32_t)kCode[seq[col + r]] << (2 * r);
            *dst = w;
          }
        }
      }
    }
  });
  tm.lap("    db: pack");
  if (!db_upload(db, words, lens, elem_group)) return nullptr;
  tm.lap("    db: upload");
  return holder.release();
}

// ---- the packed database on disk ----------------------------------------------------------
// One file per database (per shard for a node database): what sw_db_create computes from raw sequences --
// validation, the length sort, the groups of 128 lanes and the 2-bit strip stream -- stored as it goes to the
// device, so that opening it is a read plus an upload.  Sequences outside ACGT (the escape list) are stored as
// bytes; the original bytes of the packed sequences are not stored, they are unpacked from the strip stream on
// open (they are needed on the host for end cells and tracebacks of reported hits).  Little-endian, 8-byte
// aligned sections in this order: lengths in database order (u32 x n), slot -> database index (u32 x 128 x groups),
// group table, strip stream (u32 x 128 x strips), escape list (u32 x n_generic), escape bytes.
struct DbFileHeader {
  char magic[8];  // "SWHIPDB" + version byte
  uint32_t header_bytes, group_bytes;
  uint64_t n, residues, packed_residues, n_groups, n_elems, n_generic, generic_bytes;
  uint32_t max_len, reserved;
};
const char kDbMagic[8] = {'S', 'W', 'H', 'I', 'P', 'D', 'B', 1};

struct FileCloser {
  FILE* f;
  ~FileCloser() {
    if (f) fclose(f);
  }
};
bool write_section(FILE* f, const void* p, size_t bytes) {
  static const char zeros[8] = {0};
  if (bytes && fwrite(p, 1, bytes, f) != bytes) return false;
  const size_t pad = (8 - bytes % 8) % 8;
  return !pad || fwrite(zeros, 1, pad, f) == pad;
}
bool read_section(FILE* f, void* p, size_t bytes) {
  char pad[8];
  if (bytes && fread(p, 1, bytes, f) != bytes) return false;
  const size_t n = (8 - bytes % 8) % 8;
  return !n || fread(pad, 1, n, f) == n;
}

// Caller holds the context's lock.
int db_save_locked(const sw_db* db, const char* path) {
  const size_t ngroups = db->groups.size();
  std::vector<uint32_t> words((size_t)db->n_elems * kGroupTargets);
  if (!words.empty()) HIPCK(hipMemcpy(words.data(), db->d_words, words.size() * 4, hipMemcpyDeviceToHost));
  std::vector<uint32_t> lens(db->n);
  for (uint64_t i = 0; i < db->n; i++) lens[i] = (uint32_t)(db->off[i + 1] - db->off[i]);
  std::string gbytes;
  for (uint32_t ti : db->generic) gbytes.append(db->flat, db->off[ti], db->off[ti + 1] - db->off[ti]);
  DbFileHeader h{};
  memcpy(h.magic, kDbMagic, 8);
  h.header_bytes = sizeof(DbFileHeader);
  h.group_bytes = sizeof(ScoreGroup);
  h.n = db->n;
  h.residues = db->residues;
  h.packed_residues = db->packed_residues;
  h.n_groups = ngroups;
  h.n_elems = db->n_elems;
  h.n_generic = db->generic.size();
  h.generic_bytes = gbytes.size();
  h.max_len = db->max_len;
  FILE* fp = fopen(path, "wb");
  if (!fp) return fail(SW_E_ARG, std::string("sw_db_save: cannot open ") + path + ": " + strerror(errno));
  FileCloser closer{fp};
  bool ok = write_section(fp, &h, sizeof(h)) && write_section(fp, lens.data(), lens.size() * 4) &&
            write_section(fp, db->slot_target.data(), db->slot_target.size() * 4) &&
            write_section(fp, db->groups.data(), ngroups * sizeof(ScoreGroup)) &&
            write_section(fp, words.data(), words.size() * 4) &&
            write_section(fp, db->generic.data(), db->generic.size() * 4) && write_section(fp, gbytes.data(), gbytes.size());
  closer.f = nullptr;
  if (fclose(fp) != 0) ok = false;
  if (!ok) return fail(SW_E_ARG, std::string("sw_db_save: write error on ") + path);
  return SW_OK;
}

// Caller holds the context's lock (its device is current).
sw_db* db_open_locked(Context* ctx, const char* path) {
  FILE* fp = fopen(path, "rb");
  if (!fp) {
    fail(SW_E_ARG, std::string("sw_db_open: cannot open ") + path + ": " + strerror(errno));
    return nullptr;
  }
  FileCloser closer{fp};
  auto bad = [&](const char* what) -> sw_db* {
    fail(SW_E_ARG, std::string("sw_db_open: ") + path + ": " + what);
    return nullptr;
  };
  DbFileHeader h{};
  if (!read_section(fp, &h, sizeof(h))) return bad("truncated header");
  if (memcmp(h.magic, kDbMagic, 8) != 0 || h.header_bytes != sizeof(DbFileHeader) || h.group_bytes != sizeof(ScoreGroup))
    return bad("not a libsw_hip database file of this version");
  if (h.n >= 0xffffffffull || h.n_groups > h.n + 1 || h.n_elems > 0xfff00000ull || h.n_generic > h.n ||
      h.residues > ((uint64_t)1 << 46) || h.generic_bytes > h.residues || h.packed_residues > h.residues)
    return bad("implausible header");
  std::unique_ptr<sw_db, DbDeleter> holder(new sw_db);
  sw_db* db = holder.get();
  db->ctx = ctx;
  db->n = h.n;
  db->residues = h.residues;
  db->packed_residues = h.packed_residues;
  db->n_elems = (uint32_t)h.n_elems;
  db->max_len = h.max_len;
  const size_t n_slots = (size_t)h.n_groups * kGroupTargets;
  std::vector<uint32_t> lens_by_target(h.n), words((size_t)h.n_elems * kGroupTargets);
  db->slot_target.resize(n_slots);
  db->groups.resize(h.n_groups);
  db->generic.resize(h.n_generic);
  std::string gbytes(h.generic_bytes, '\0');
  if (!read_section(fp, lens_by_target.data(), lens_by_target.size() * 4) ||
      !read_section(fp, db->slot_target.data(), n_slots * 4) ||
      !read_section(fp, db->groups.data(), db->groups.size() * sizeof(ScoreGroup)) ||
      !read_section(fp, words.data(), words.size() * 4) || !read_section(fp, db->generic.data(), db->generic.size() * 4) ||
      !read_section(fp, &gbytes[0], gbytes.size()))
    return bad("truncated file");
  // the kernels index with these tables: check every one against the others before anything reaches the device
  db->off.assign(h.n + 1, 0);
  for (uint64_t i = 0; i < h.n; i++) {
    if (lens_by_target[i] >= 0x7fffff00u || lens_by_target[i] > h.max_len) return bad("sequence length out of range");
    db->off[i + 1] = db->off[i] + lens_by_target[i];
  }
  if (db->off[h.n] != h.residues) return bad("lengths do not add up to the residue count");
  uint64_t strips = 0;
  for (size_t g = 0; g < db->groups.size(); g++) {
    const ScoreGroup& G = db->groups[g];
    if (G.strip_base != strips || G.first_slot != g * kGroupTargets || G.nstrips == 0) return bad("inconsistent group table");
    strips += G.nstrips;
  }
  if (strips != h.n_elems) return bad("group table and strip stream disagree");
  db->packed_ok.assign(h.n, 0);
  std::vector<uint32_t> lens(n_slots, 0);
  uint64_t packed_res = 0;
  for (size_t k = 0; k < n_slots; k++) {
    const uint32_t ti = db->slot_target[k];
    if (ti == UINT32_MAX) continue;
    if (ti >= h.n || db->packed_ok[ti] || lens_by_target[ti] == 0) return bad("slot table names a bad sequence");
    if ((lens_by_target[ti] + kScoreR - 1) / kScoreR > db->groups[k / kGroupTargets].nstrips) return bad("sequence longer than its group");
    db->packed_ok[ti] = 1;
    lens[k] = lens_by_target[ti];
    packed_res += lens[k];
  }
  if (packed_res != h.packed_residues) return bad("packed residue count mismatch");
  // derived from the lengths, never taken from the file: the kernel picks its half-width row loop by it
  for (size_t g = 0; g < db->groups.size(); g++) {
    uint32_t maxlen = 0;
    for (size_t k = g * kGroupTargets; k < (g + 1) * kGroupTargets && k < n_slots; k++) maxlen = std::max(maxlen, lens[k]);
    ScoreGroup& G = db->groups[g];
    const uint64_t before_last = (uint64_t)kScoreR * (G.nstrips - 1);
    G.last_cols = maxlen > before_last ? (uint32_t)(maxlen - before_last) : 0;
  }
  count_half_strips(db, lens);
  uint64_t gb = 0;
  for (uint32_t ti : db->generic) {
    if (ti >= h.n || db->packed_ok[ti] == 1 || db->packed_ok[ti] == 2) return bad("escape list names a bad sequence");
    db->packed_ok[ti] = 2;
    gb += lens_by_target[ti];
  }
  if (gb != h.generic_bytes || packed_res + gb != h.residues) return bad("escape bytes mismatch");
  for (uint64_t i = 0; i < h.n; i++)
    if (!db->packed_ok[i]) return bad("a sequence is neither packed nor on the escape list");
  // rebuild the host copy of the sequences: escape bytes as stored, packed ones out of the strip stream
  db->flat.assign(h.residues, '\0');
  gb = 0;
  for (uint32_t ti : db->generic) {
    if (memchr(gbytes.data() + gb, 0, lens_by_target[ti])) return bad("NUL byte inside a sequence");
    Decoded d;
    if (!decode(gbytes.data() + gb, lens_by_target[ti], d)) return bad("escape sequence is not valid UTF-8");
    memcpy(&db->flat[db->off[ti]], gbytes.data() + gb, lens_by_target[ti]);
    db->packed_ok[ti] = 0;
    gb += lens_by_target[ti];
  }
  std::vector<uint32_t> elem_group(std::max<size_t>(h.n_elems, 1), 0);
  static const char kNt[4] = {'A', 'C', 'G', 'T'};
  parallel_ranges(db->groups.size(), db->groups.size() >= 32 ? (size_t)h.residues + ((size_t)1 << 20) : 0, [&](size_t glo, size_t ghi) {
    for (size_t g = glo; g < ghi; g++) {
      const ScoreGroup& G = db->groups[g];
      for (uint32_t st = 0; st < G.nstrips; st++) elem_group[G.strip_base + st] = (uint32_t)g;
      for (int sl = 0; sl < kGroupTargets; sl++) {
        const size_t k = g * kGroupTargets + sl;
        const uint32_t ti = db->slot_target[k];
        if (ti == UINT32_MAX) continue;
        char* dst = &db->flat[db->off[ti]];
        const uint32_t len = lens[k];
        const uint32_t* src = words.data() + (size_t)G.strip_base * kGroupTargets + (size_t)(sl / 64) * 64 + (sl % 64);
        for (uint32_t col = 0; col < len; col += kScoreR, src += kGroupTargets) {
          uint32_t w = *src;
          const uint32_t nr = std::min<uint32_t>(kScoreR, len - col);
          for (uint32_t r = 0; r < nr; r++, w >>= 2) dst[col + r] = kNt[w & 3];
        }
      }
    }
  });
  if (words.empty()) words.push_back(0);
  if (!db_upload(db, words, lens, elem_group)) return nullptr;
  return holder.release();
}

// Ticket order of the score kernel's work units.  The strip stream is cut into lanes (runs of whole groups,
// cut where the last unit is nearly full); S lanes are active at a time and tickets visit them round-robin, so
// consecutive units of one lane -- the only ones that depend on each other -- are S tickets apart.
// R = slots of the boundary ring (slot of ticket t: t % R); `reuse` = the latest earlier reader of that slot.
//
// The tail: n_units is rarely a multiple of the number of persistent workgroups, and the last, partial round would
// keep a handful of CUs busy for a whole unit time while the rest idle (bench shape: 3079 units on 256 CUs = 12
// rounds + 7 units, i.e. 8 % of the kernel for 0.2 % of the work).  The units of that partial round are cut into
// up to four short units (fewer strips): they land on more CUs with one wave per SIMD instead of four, which runs
// a strip more than twice as fast.
// The core works on caller-provided arrays and allocates nothing (sw_db_scores_noalloc runs it inside a
// pre-reserved arena); capacities: lanes >= n_groups, active >= S, slot_reader >= R, units >= units_bound().
struct UnitLane { uint32_t start, pos, end, last_ticket; };
size_t units_bound(uint32_t n_elems, size_t n_groups, uint32_t W, uint32_t max_wg) {
  return (size_t)n_elems / W + n_groups + 3 * (size_t)max_wg + 4;
}
// returns the number of units, or -1 if `units_cap` is too small (never with units_bound())
int64_t build_units_raw(const ScoreGroup* groups, size_t ng, uint32_t n_elems, uint32_t W, uint32_t S, uint32_t R,
                        uint32_t max_wg, ScoreUnit* units, size_t units_cap, UnitLane* lanes, size_t* active,
                        uint32_t* slot_reader, uint64_t* chained_out) {
  size_t n_lanes = 0;
  const uint64_t approx_units = ((uint64_t)n_elems + W - 1) / W;
  const uint64_t Lmin = std::min<uint64_t>(32, std::max<uint64_t>(2, approx_units / (4ull * S)));
  // A lane is a chain of dependent units: with fewer lanes than resident workgroups a unit's predecessor is still
  // running when the unit starts, and its wave 0 waits for the first published blocks at every unit.  Lanes are
  // therefore no longer than it takes to have about one lane per workgroup; within that, the cut that fills the
  // last unit best wins.  (Measured, 10 kb query: 65-strip groups 80.1 -> 78.4 ms, 33-strip groups 83.8 -> 80.8 ms;
  // lanes of 16 groups of 63 strips, which would fill every unit, cost +8 %.)
  uint64_t Lmax = std::min<uint64_t>(4 * Lmin + 16, std::max<uint64_t>(Lmin + 2, approx_units / std::max<uint32_t>(max_wg, 1)));
  if (const char* e = getenv("SW_HIP_SCORE_LMAX")) Lmax = std::max(1, atoi(e));  // experiments
  for (size_t g = 0; g < ng;) {
    uint64_t cum = 0;
    size_t best = g;
    double best_fill = -1;
    for (size_t h = g; h < ng; h++) {
      cum += groups[h].nstrips;
      const uint64_t u = (cum + W - 1) / W;
      if (u < Lmin && h + 1 < ng) continue;
      const double fill = (double)cum / (double)(u * W);
      if (fill > best_fill) best_fill = fill, best = h;
      if (fill >= 0.995 || u >= Lmax) break;
    }
    lanes[n_lanes++] = UnitLane{groups[g].strip_base, groups[g].strip_base, groups[best].strip_base + groups[best].nstrips, UINT32_MAX};
    g = best + 1;
  }
  // pass 0 counts the units; pass 1 generates them, with short units from ticket `full` on
  uint32_t full = UINT32_MAX, Wtail = W;
  size_t n_units = 0;
  uint64_t chained = 0;
  for (int pass = 0; pass < 3; pass++) {
    for (size_t l = 0; l < n_lanes; l++) lanes[l].pos = lanes[l].start, lanes[l].last_ticket = UINT32_MAX;
    n_units = 0;
    chained = 0;
    for (uint32_t r = 0; r < R; r++) slot_reader[r] = UINT32_MAX;
    size_t n_active = 0, next_lane = 0;
    while (n_active < S && next_lane < n_lanes) active[n_active++] = next_lane++;
    while (n_active) {
      for (size_t i = 0; i < n_active;) {
        UnitLane& L = lanes[active[i]];
        if (n_units >= units_cap) return -1;
        const uint32_t t = (uint32_t)n_units;
        ScoreUnit u;
        u.first_elem = L.pos;
        u.count = std::min<uint32_t>(t >= full ? Wtail : W, L.end - L.pos);
        u.dep = L.last_ticket;
        u.reuse = slot_reader[t % R];
        if (u.dep != UINT32_MAX) slot_reader[u.dep % R] = t;
        units[n_units++] = u;
        L.pos += u.count;
        L.last_ticket = t;
        if (L.pos >= L.end) {
          if (next_lane < n_lanes) {
            active[i++] = next_lane++;
          } else {
            for (size_t k = i; k + 1 < n_active; k++) active[k] = active[k + 1];
            n_active--;
          }
        } else {
          i++;
        }
      }
    }
    if (pass == 0) {
      const uint32_t n = (uint32_t)n_units;
      const uint32_t r = max_wg ? n % max_wg : 0;
      if (n <= max_wg || r == 0 || W < 8) break;  // no partial round worth cutting
      const uint32_t f = std::min<uint32_t>(4, max_wg / r);
      if (f >= 2) {
        Wtail = W / f;
      } else {
        // more than half a round left: units of 3/4 the strips (3 waves per SIMD instead of 4) if they still fit
        // one round -- every workgroup then works for about 3/4 of a unit time instead of some for a whole one
        const uint32_t c = (uint32_t)((((uint64_t)r * W + max_wg - 1) / max_wg + 3) & ~(uint64_t)3);
        if (c >= W) break;
        Wtail = c;
      }
      full = n - r;
    } else if (pass == 1) {
      // the short units must fit ONE round (lane ends add partial units): otherwise back to full units
      if (full == UINT32_MAX || n_units - full <= max_wg) break;
      full = UINT32_MAX;
      Wtail = W;
    }
  }
  // units whose last strip is not the last strip of its group hand their H column to the next unit through HBM
  if (chained_out) {
    size_t g = 0;
    for (size_t t = 0; t < n_units; t++) {  // (units of one lane are in stream order but lanes interleave: search per unit)
      const uint32_t last = units[t].first_elem + units[t].count - 1;
      size_t lo = 0, hi = ng;
      while (lo + 1 < hi) {
        const size_t mid = (lo + hi) / 2;
        if (groups[mid].strip_base <= last) lo = mid;
        else hi = mid;
      }
      g = lo;
      if (last + 1 < groups[g].strip_base + groups[g].nstrips) chained++;
    }
    *chained_out = chained;
  }
  return (int64_t)n_units;
}

void build_units(const sw_db* db, uint32_t W, uint32_t S, uint32_t R, uint32_t max_wg, std::vector<ScoreUnit>& units,
                 uint64_t* chained_out = nullptr) {
  const size_t ng = db->groups.size();
  std::vector<UnitLane> lanes(ng + 1);
  std::vector<size_t> active(S + 1);
  std::vector<uint32_t> slot_reader(R);
  units.resize(units_bound(db->n_elems, ng, W, max_wg));
  const int64_t n = build_units_raw(db->groups.data(), ng, db->n_elems, W, S, R, max_wg, units.data(), units.size(),
                                    lanes.data(), active.data(), slot_reader.data(), chained_out);
  units.resize(n < 0 ? 0 : (size_t)n);
}

}  // namespace
// Test hook (host arithmetic only, no GPU): builds the ticket-ordered unit table for a database whose groups have
// the given numbers of strips and checks the invariants the score kernel relies on -- every strip is covered
// exactly once, units are 1..W strips, a unit that continues a group depends on the EARLIER ticket that ends right
// before it, the ring slot it writes was last read by an earlier ticket.  Returns the number of units, or a
// negative number naming the violated invariant.  out[0] = units of the short (tail) kind, out[1] = their size.
extern "C" int64_t sw_hip_test_units(const uint32_t* group_strips, uint32_t n_groups, uint32_t W, uint32_t S, uint32_t R,
                                     uint32_t max_wg, uint32_t* out) {
  if (!group_strips || !W || !S || !R) return -100;
  return swhip::guarded<int64_t>(-101, [&]() -> int64_t {
  sw_db db;
  uint32_t total = 0;
  std::vector<uint32_t> elem_group;
  for (uint32_t g = 0; g < n_groups; g++) {
    ScoreGroup G{};
    G.strip_base = total;
    G.nstrips = group_strips[g];
    G.first_slot = g * kGroupTargets;
    db.groups.push_back(G);
    for (uint32_t k = 0; k < group_strips[g]; k++) elem_group.push_back(g);
    total += group_strips[g];
  }
  db.n_elems = total;
  std::vector<ScoreUnit> units;
  build_units(&db, W, S, R, max_wg, units);
  std::vector<uint8_t> covered(total, 0);
  std::vector<uint32_t> last_reader(R, UINT32_MAX);
  uint32_t n_short = 0, short_size = 0;
  for (uint32_t t = 0; t < units.size(); t++) {
    const ScoreUnit& u = units[t];
    if (u.count < 1 || u.count > W) return -1;
    if ((uint64_t)u.first_elem + u.count > total) return -2;
    for (uint32_t k = 0; k < u.count; k++) {
      if (covered[u.first_elem + k]) return -3;
      covered[u.first_elem + k] = 1;
    }
    const uint32_t g0 = elem_group[u.first_elem];
    const bool continues = u.first_elem > db.groups[g0].strip_base;  // wave 0 needs a left column from another unit
    if (continues) {
      if (u.dep == UINT32_MAX || u.dep >= t) return -4;
      const ScoreUnit& d = units[u.dep];
      if (d.first_elem + d.count != u.first_elem) return -5;
    }
    if (u.reuse != UINT32_MAX && u.reuse >= t) return -6;
    if (u.reuse != last_reader[t % R]) return -7;
    if (u.dep != UINT32_MAX) {
      if (u.dep >= t) return -8;
      last_reader[u.dep % R] = t;
    }
    if (u.count < W && u.first_elem + u.count < total &&
        elem_group[u.first_elem + u.count] == elem_group[u.first_elem + u.count - 1]) {  // short, and the group goes on
      n_short++;
      short_size = std::max(short_size, u.count);
    }
  }
  for (uint32_t e = 0; e < total; e++)
    if (!covered[e]) return -9;
  if (out) {
    out[0] = n_short;
    out[1] = short_size;
  }
  return (int64_t)units.size();
  });
}

namespace {

bool fast_domain(Scoring sc) {
  return sc.match > 0 && sc.miss < 0 && sc.gap < 0 && sc.match <= 16384 && sc.miss >= -16384 && sc.gap >= -16384;
}

// Launches the score kernel for one query over the packed part of the database (asynchronous, on the library's
// stream; ev0/ev1 bracket the kernel).  The per-slot maxima are left in ws.slot_scores.
// `arena` != nullptr: the allocation-free variant (sw_db_scores_noalloc) -- ws is the fixed workspace carved out
// of the device's static scratch array, the unit table is built inside the pre-reserved host arena on every call
// and the number of workgroups is cut down to what the fixed boundary ring can serve.
// A batch of queries in one pass (ScoreParams::blkinfo): `batch` names the row blocks of the stacked queries, qwords
// holds one word per block, Q is the length of one of them and ws.slot_scores receives one array per query.
struct ScoreBatch {
  const uint32_t* blkinfo;  // n_blocks + 1 words
  uint32_t n_blocks, n_queries;
  uint64_t rows;            // sum of the query lengths
};
// qwords_small: the same query packed kScoreBSmall rows per word (QueryPrep::qwords_small); given it, a pass over a
// small database runs with the small hand-over block (score_block_rows).  *block_rows_out: what the pass ran with.
int score_pass_locked(Context& c, Workspace& ws, sw_db* db, const uint32_t* qwords, size_t n_qwords, uint32_t Q,
                      Scoring sc, HostArena* arena = nullptr, bool persist = false, const ScoreBatch* batch = nullptr,
                      hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr,
                      const std::vector<uint32_t>* qwords_small = nullptr, int* block_rows_out = nullptr) {
  if (batch && (arena || persist || batch->n_blocks == 0 || n_qwords != (size_t)batch->n_blocks + 1))
    return fail(SW_E_ARG, "internal: bad query batch");
  const uint32_t W = (uint32_t)score_waves_per_group(db->n_elems, c.cus, (uint32_t)db->groups.size());
  int BR = kScoreB;
  if (!arena && !batch && qwords_small && score_block_rows((int)W, false, (uint32_t)db->groups.size()) == kScoreBSmall) {
    BR = kScoreBSmall;
    qwords = qwords_small->data();
    n_qwords = qwords_small->size();
    if (n_qwords != (Q + (uint32_t)BR - 1) / (uint32_t)BR + 1) return fail(SW_E_ARG, "internal: bad query words");
  }
  if (block_rows_out) *block_rows_out = BR;
  const uint32_t nblk = batch ? batch->n_blocks : (Q + (uint32_t)BR - 1) / (uint32_t)BR;
  const uint32_t n_sets = batch ? batch->n_queries : 1;
  const size_t Qpad = (size_t)nblk * (size_t)BR;
  uint32_t max_wg = (uint32_t)c.cus * (uint32_t)score_workgroups_per_cu((int)W);
  // Active lanes S of the ticket order (build_units): the unit that continues a group is handed out S tickets
  // after its producer.  With S beyond the number of resident workgroups the producer has FINISHED by then and the
  // consumer never waits; a consumer that trails a running producer (small S) costs 5-9 % whatever the lag
  // (swept 1..8192 on every shape of scratch/sweep.py: flat from ~512 up).  So: as many lanes as the database has
  // groups, up to 1024, as long as the boundary ring (one slot of Qpad x 256 B per unit in flight or waiting) stays
  // within twice the scratch budget; a database with few groups or a very long query falls back to chains.
  const uint32_t nblk16 = batch ? (uint32_t)((Qpad + 15) / 16) : (Q + 15) / 16;
  // boundary ring: a workgroup has at most 1 + queue depth units in flight, fewer when a unit is long
  const uint32_t ahead = std::min<uint32_t>(score_queue_depth(), 48 / nblk16 + 1);
  uint32_t S, ring_slots;
  uint32_t n_units;
  uint64_t chained = 0;
  const ScoreUnit* d_units;
  if (!arena) {
    const uint32_t ring_min = max_wg * (1 + ahead) + 8;
    const uint64_t ring_fit = 2 * (uint64_t)c.scratch_budget / ((uint64_t)Qpad * 256);
    S = (uint32_t)std::min<uint64_t>(1024, std::max<uint64_t>(8, db->groups.size()));
    if (ring_fit < (uint64_t)ring_min + S) S = (uint32_t)std::max<uint64_t>(8, ring_fit > ring_min ? ring_fit - ring_min : 0);
    if (const char* e = getenv("SW_HIP_SCORE_LANES")) S = std::min(16384, std::max(1, atoi(e)));  // experiments
    ring_slots = S + max_wg * (1 + ahead) + 8;
    // persist: every unit keeps a ring slot of its own (slot = ticket), so the boundary columns between units
    // survive the launch -- the work-package fast path re-enters the matrix at them (sw_package.cpp)
    if (persist) ring_slots = (uint32_t)std::min<size_t>(0x7fffffffu, units_bound(db->n_elems, db->groups.size(), W, max_wg) + 1);
    if (db->units_S != S || db->units_R != ring_slots || db->units_W != W || !db->d_units) {
      std::vector<ScoreUnit> units;
      build_units(db, W, S, ring_slots, max_wg, units, &db->units_chained);
      db->h_units = std::make_shared<const std::vector<ScoreUnit>>(units);
      if (db->borrowed) {
        // (a pass of the previous geometry may still be reading the table: hipFree below waits for the whole device,
        // here the context's stream is enough)
        HIPCK(hipStreamSynchronize(c.stream));
        int erc;
        if ((erc = c.tdb.units.ensure(std::max<size_t>(units.size(), 1) * sizeof(ScoreUnit)))) return erc;
        db->d_units = (ScoreUnit*)c.tdb.units.p;
        HIPCK(hipMemcpyAsync(db->d_units, units.data(), units.size() * sizeof(ScoreUnit), hipMemcpyHostToDevice, c.stream));
        HIPCK(hipStreamSynchronize(c.stream));
      } else {
        if (db->d_units) (void)hipFree(db->d_units);
        db->d_units = nullptr;
        if (hipMalloc(&db->d_units, std::max<size_t>(units.size(), 1) * sizeof(ScoreUnit)) != hipSuccess)
          return fail(SW_E_NOMEM, "hipMalloc failed for the unit table");
        HIPCK(hipMemcpy(db->d_units, units.data(), units.size() * sizeof(ScoreUnit), hipMemcpyHostToDevice));
      }
      db->n_units = (uint32_t)units.size();
      db->units_S = S;
      db->units_R = ring_slots;
      db->units_W = W;
    }
    n_units = db->n_units;
    chained = db->units_chained;
    d_units = db->d_units;
  } else {
    // the fixed ring decides how many workgroups may run: ring_slots = S + nwg * (1 + ahead) + 8 slots of Qpad x 256 B
    const uint64_t ring_fit = ws.gbound.cap / ((uint64_t)Qpad * 256);
    S = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(2, db->groups.size()));
    if (ring_fit < (uint64_t)S + 8 + (1 + ahead)) S = 2;
    if (ring_fit < (uint64_t)S + 8 + (1 + ahead))
      return fail(SW_E_LIMIT, "sw_db_scores_noalloc: the query is too long for the fixed boundary ring (" +
                                  std::to_string(ws.gbound.cap >> 20) + " MiB hold " + std::to_string(ring_fit) +
                                  " columns of this query, " + std::to_string(S + 8 + 1 + ahead) + " are needed)");
    max_wg = (uint32_t)std::min<uint64_t>(max_wg, (ring_fit - S - 8) / (1 + ahead));
    ring_slots = S + max_wg * (1 + ahead) + 8;
    const size_t ng = db->groups.size();
    const size_t ucap = units_bound(db->n_elems, ng, W, max_wg);
    ScoreUnit* units = (ScoreUnit*)arena->take(ucap * sizeof(ScoreUnit));
    UnitLane* lanes = (UnitLane*)arena->take((ng + 1) * sizeof(UnitLane));
    size_t* active = (size_t*)arena->take(((size_t)S + 1) * sizeof(size_t));
    uint32_t* slot_reader = (uint32_t*)arena->take((size_t)ring_slots * 4);
    if (!units || !lanes || !active || !slot_reader)
      return fail(SW_E_LIMIT, "sw_db_scores_noalloc: the database has too many groups for the pre-reserved host arena");
    const int64_t n = build_units_raw(db->groups.data(), ng, db->n_elems, W, S, ring_slots, max_wg, units, ucap, lanes,
                                      active, slot_reader, &chained);
    if (n < 0) return fail(SW_E_LIMIT, "sw_db_scores_noalloc: unit table overflow");
    n_units = (uint32_t)n;
    int rc0;
    if ((rc0 = ws.units.ensure(std::max<size_t>((size_t)n_units, 1) * sizeof(ScoreUnit)))) return rc0;
    HIPCK(hipMemcpyAsync(ws.units.p, units, (size_t)n_units * sizeof(ScoreUnit), hipMemcpyHostToDevice, c.stream));
    d_units = (const ScoreUnit*)ws.units.p;
  }
  const int nwg = (int)std::min<uint32_t>(n_units, max_wg);
  const size_t gb_stride = Qpad * 64;
  int rc;
  if ((rc = ws.qwords.ensure(n_qwords * 4))) return rc;
  if ((rc = ws.slot_scores.ensure(db->slot_target.size() * 4 * n_sets))) return rc;
  if (batch && (rc = ws.blkinfo.ensure(((size_t)nblk + 1) * 4))) return rc;
  if ((rc = ws.gbound.ensure((size_t)(persist ? n_units : ring_slots) * gb_stride * 4))) return rc;
  if (persist && (rc = ws.strip_best.ensure(std::max<size_t>((size_t)db->n_elems * 256, 256)))) return rc;
  if ((rc = ws.ticket.ensure(256))) return rc;
  if ((rc = ws.prog.ensure(std::max<size_t>((size_t)n_units * 8, 16)))) return rc;
  HIPCK(hipMemcpyAsync(ws.qwords.p, qwords, n_qwords * 4, hipMemcpyHostToDevice, c.stream));
  if (batch) HIPCK(hipMemcpyAsync(ws.blkinfo.p, batch->blkinfo, ((size_t)nblk + 1) * 4, hipMemcpyHostToDevice, c.stream));
  HIPCK(hipMemsetAsync(ws.ticket.p, 0, 256, c.stream));
  HIPCK(hipMemsetAsync(ws.prog.p, 0, std::max<size_t>((size_t)n_units * 8, 16), c.stream));
  HIPCK(hipMemsetAsync(ws.slot_scores.p, 0, db->slot_target.size() * 4 * n_sets, c.stream));
  ScoreParams p{};
  p.words = db->d_words;
  p.groups = db->d_groups;
  p.lens = db->d_lens;
  p.qwords = (const uint32_t*)ws.qwords.p;
  p.Q = Q;
  p.elem_group = db->d_elem_group;
  p.units = d_units;
  p.n_units = n_units;
  p.n_elems = db->n_elems;
  p.n_groups = (uint32_t)db->groups.size();
  p.n_slots = (uint32_t)db->slot_target.size();
  p.slot_scores = (int32_t*)ws.slot_scores.p;
  p.gbound = (uint32_t*)ws.gbound.p;
  p.gbound_stride = gb_stride;
  p.ring_slots = ring_slots;
  p.strip_best = persist ? (uint32_t*)ws.strip_best.p : nullptr;
  p.prog = (uint32_t*)ws.prog.p;
  p.ticket = (uint32_t*)ws.ticket.p;
  p.err = (uint32_t*)ws.ticket.p + 32;
  p.match = sc.match;
  p.miss = sc.miss;
  p.gap = sc.gap;
  p.blkinfo = batch ? (const uint32_t*)ws.blkinfo.p : nullptr;
  p.n_blocks = batch ? nblk : 0;
  // a small database is a few chains of units: publish their boundary columns often (latency), a big one rarely
  // (in blocks; swept: 10 kb queries like 10 blocks, 1 kb queries 5 -- a short unit is over before a coarse
  // publication lets its consumer start)
  p.pub = W == 4 ? 2 : nblk >= 400 ? 10 : 5;
  if (const char* e = getenv("SW_HIP_SCORE_PUB")) p.pub = std::max(1, atoi(e));  // experiments
  // (the kernel's bracket: the context's pair of events, or the caller's when several passes are in flight)
  HIPCK(hipEventRecord(ev_begin ? ev_begin : c.ev0, c.stream));
  HIPCK(launch_score(p, nwg, (int)W, BR, c.stream));
  HIPCK(hipEventRecord(ev_end ? ev_end : c.ev1, c.stream));
  // what this launch executes, for the roofline arithmetic of bench.py (counted, not pasted: every strip of the
  // stream walks all Q query rows through the generated row loop; a unit that hands its last column to the next
  // unit also stores those Q rows to HBM, 256 bytes each, and the consumer reads them back once)
  // (in full-strip equivalents: a half-width strip executes half the packed instructions per row)
  const uint64_t rows = batch ? batch->rows : Q;
  c.stats.score_strip_rows += ((uint64_t)2 * db->n_elems - db->n_half_elems) * rows / 2;
  c.stats.score_chain_rows += chained * rows;
  c.stats.score_workgroups = (uint32_t)nwg;
  return SW_OK;
}

// after the stream has been synchronised: error word + kernel time of the last score pass
int score_pass_finish_locked(Context& c, sw_db* db, uint64_t Q, uint32_t kerr, hipEvent_t ev_begin = nullptr,
                             hipEvent_t ev_end = nullptr) {  // Q: query rows of the pass (summed over a batch)
  if (kerr) {
    static const char* const what[] = {"", "a wait for another workgroup's boundary column ran out of time",
                                       "a wait for a neighbouring wave ran out of time",
                                       "the unit queue of a workgroup stalled", "inconsistent unit table entry",
                                       "boundary ring slot smaller than the query", "strip with a bad group index",
                                       "strip outside its group / chained unit without a producer"};
    return fail(SW_E_HIP, std::string("score kernel: ") + (kerr < 8 ? what[kerr] : "unknown error") + " (code " +
                              std::to_string(kerr) + ")");
  }
  float ms = 0;
  HIPCK(hipEventElapsedTime(&ms, ev_begin ? ev_begin : c.ev0, ev_end ? ev_end : c.ev1));
  c.stats.score_kernel_ms += ms;
  c.stats.score_launches += 1;
  c.stats.score_cells += Q * db->packed_residues;
  return SW_OK;
}

// A packed target's int16 score is final unless one of these holds (the same rules run on the
// device in sw_topk.hip):
bool needs_exact(uint32_t T, int32_t s, uint32_t Q, uint32_t Qp, Scoring sc) {
  const bool full = (uint64_t)T < 2ull * Q;
  if (s >= 32767) return true;                                                    // int16 saturation -> int32 re-run
  if (!full && T < Qp) return true;                                               // ring path with T < Qp (quirk Q2)
  if (full && T % 64 == 0 && roundup64(T) >= Qp && s < sc.match) return true;     // only H(1,T) can score
  return false;
}

// The database path answers what the WORKER would have got for the pair, i.e. the fallback chain of
// worker_impl.go:170-202: find_alignment_low_memory, or find_alignment_simd where that one panics (gap == 0 on
// the ring path, rust/src/algorithm/mod.rs:236-238).
int db_pair_path(uint32_t Q, uint32_t T, Scoring sc) {
  return ring_div_by_zero(SW_PATH_LOW_MEMORY, Q, T, sc) ? SW_PATH_SIMD : SW_PATH_LOW_MEMORY;
}

// exact (int32, score only) re-run of a list of database targets
int rescore_locked(Context& c, Workspace& ws, sw_db* db, const QueryPrep& qp, Scoring sc,
                   const std::vector<uint32_t>& rerun, std::vector<int32_t>& out) {
  out.assign(rerun.size(), 0);
  if (rerun.empty()) return SW_OK;
  const uint32_t Q = (uint32_t)qp.d.sym.size();
  std::vector<PairSpec> specs(rerun.size());
  std::vector<Decoded> tdec(rerun.size());
  for (size_t k = 0; k < rerun.size(); k++) {
    uint32_t ti = rerun[k];
    // (valid UTF-8: checked when the database was built)
    decode(db->flat.data() + db->off[ti], (size_t)(db->off[ti + 1] - db->off[ti]), tdec[k]);
    specs[k] = PairSpec{qp.d.sym.data(), Q, tdec[k].sym.data(), (uint32_t)tdec[k].sym.size(),
                        db_pair_path(Q, (uint32_t)tdec[k].sym.size(), sc)};
  }
  std::vector<PairOut> po(rerun.size());
  int rc = run_exact(c, ws, specs.data(), specs.size(), sc, false, po.data());
  if (rc) return rc;
  for (size_t k = 0; k < rerun.size(); k++) out[k] = po[k].score;
  c.stats.rerun_pairs += (uint32_t)rerun.size();
  return SW_OK;
}

bool fast_path(const sw_db* db, const QueryPrep& qp, Scoring sc) {
  return qp.acgt && fast_domain(sc) && !qp.d.sym.empty() && !db->groups.empty();
}

// scores of every database sequence against one query, reference-exact (low_memory semantics)
int db_scores_locked(Context& c, Workspace& ws, sw_db* db, const QueryPrep& qp, Scoring sc, std::vector<int32_t>& scores) {
  StreamDrain drain(c.stream);
  const uint32_t Q = (uint32_t)qp.d.sym.size();
  scores.assign(db->n, 0);
  std::vector<uint32_t> rerun(db->generic);
  if (!fast_path(db, qp, sc)) {
    rerun.resize(db->n);
    std::iota(rerun.begin(), rerun.end(), 0u);
  } else {
    int rc = score_pass_locked(c, ws, db, qp.qwords.data(), qp.qwords.size(), (uint32_t)qp.d.sym.size(), sc, nullptr, false,
                               nullptr, nullptr, nullptr, &qp.qwords_small);
    if (rc) return rc;
    std::vector<int32_t> slot(db->slot_target.size());
    HIPCK(hipMemcpyAsync(slot.data(), ws.slot_scores.p, slot.size() * 4, hipMemcpyDeviceToHost, c.stream));
    uint32_t kerr = 0;
    HIPCK(hipMemcpyAsync(&kerr, (uint32_t*)ws.ticket.p + 32, 4, hipMemcpyDeviceToHost, c.stream));
    HIPCK(hipStreamSynchronize(c.stream));
    if ((rc = score_pass_finish_locked(c, db, Q, kerr))) return rc;
    const uint32_t Qp = roundup64(Q);
    for (size_t k = 0; k < slot.size(); k++) {
      uint32_t ti = db->slot_target[k];
      if (ti == UINT32_MAX) continue;
      const uint32_t T = (uint32_t)(db->off[ti + 1] - db->off[ti]);
      if (needs_exact(T, slot[k], Q, Qp, sc)) rerun.push_back(ti);
      else scores[ti] = slot[k];
    }
  }
  std::vector<int32_t> rs;
  int rc = rescore_locked(c, ws, db, qp, sc, rerun, rs);
  if (rc) return rc;
  for (size_t k = 0; k < rerun.size(); k++) scores[rerun[k]] = rs[k];
  return SW_OK;
}

bool g_host_topk = [] {
  const char* e = getenv("SW_HIP_TOPK");
  return e && std::string(e) == "host";
}();
constexpr uint32_t kTieCap = 65536;

// Device-side selection (sw_topk.hip) over the per-slot scores of ONE query that a score pass left at d_slot: the
// candidates above the k-th score, the ties at it and the slots that need an exact re-run come back, the re-runs
// are done, and everything that can be among the k best lands in `pool`.  pass_rows != 0: this is the first
// selection after the score pass -- its error word is fetched with the same synchronisation and the pass is
// accounted (pass_rows = query rows of the pass).  done = false: the tie list overflowed (e.g. every target scores
// the same); the caller takes the full download instead.
int select_pool_locked(Context& c, Workspace& ws, sw_db* db, const QueryPrep& qp, Scoring sc, size_t k,
                       const int32_t* d_slot, uint64_t pass_rows, std::vector<TopEntry>& pool, bool& done) {
  done = false;
  const uint32_t Q = (uint32_t)qp.d.sym.size();
  const uint32_t n_slots = (uint32_t)db->slot_target.size();
  int rc;
  if ((rc = ws.sel_hist.ensure(32768 * 4))) return rc;
  if ((rc = ws.sel_ctr.ensure(kSelWords * 4))) return rc;
  if ((rc = ws.sel_rerun.ensure((size_t)n_slots * 4))) return rc;
  if ((rc = ws.sel_cand.ensure(std::max<size_t>(k, 16) * 8))) return rc;
  if ((rc = ws.sel_tie.ensure((size_t)kTieCap * 8))) return rc;
  HIPCK(hipMemsetAsync(ws.sel_hist.p, 0, 32768 * 4, c.stream));
  HIPCK(hipMemsetAsync(ws.sel_ctr.p, 0, kSelWords * 4, c.stream));
  SelectParams sp{};
  sp.slot_scores = d_slot;
  sp.lens = db->d_lens;
  sp.slot_target = db->d_slot_target;
  sp.n_slots = n_slots;
  sp.Q = Q;
  sp.match = sc.match;
  sp.k = (uint32_t)k;
  sp.hist = (uint32_t*)ws.sel_hist.p;
  sp.counters = (uint32_t*)ws.sel_ctr.p;
  sp.rerun = (uint32_t*)ws.sel_rerun.p;
  sp.rerun_cap = n_slots;
  sp.cand = (uint32_t*)ws.sel_cand.p;
  sp.cand_cap = (uint32_t)std::max<size_t>(k, 16);
  sp.tie = (uint32_t*)ws.sel_tie.p;
  sp.tie_cap = kTieCap;
  HIPCK(launch_select(sp, c.stream));
  uint32_t ctr[kSelWords] = {0};
  uint32_t kerr = 0;
  HIPCK(hipMemcpyAsync(ctr, ws.sel_ctr.p, sizeof(ctr), hipMemcpyDeviceToHost, c.stream));
  if (pass_rows) HIPCK(hipMemcpyAsync(&kerr, (uint32_t*)ws.ticket.p + 32, 4, hipMemcpyDeviceToHost, c.stream));
  HIPCK(hipStreamSynchronize(c.stream));
  if (pass_rows && (rc = score_pass_finish_locked(c, db, pass_rows, kerr))) return rc;
  if (!(ctr[kSelTie] <= kTieCap && ctr[kSelCand] <= sp.cand_cap && ctr[kSelRerun] <= n_slots)) return SW_OK;
  std::vector<uint32_t> cand(2 * (size_t)ctr[kSelCand]), tie(2 * (size_t)ctr[kSelTie]), rr(ctr[kSelRerun]);
  if (!cand.empty()) HIPCK(hipMemcpyAsync(cand.data(), ws.sel_cand.p, cand.size() * 4, hipMemcpyDeviceToHost, c.stream));
  if (!tie.empty()) HIPCK(hipMemcpyAsync(tie.data(), ws.sel_tie.p, tie.size() * 4, hipMemcpyDeviceToHost, c.stream));
  if (!rr.empty()) HIPCK(hipMemcpyAsync(rr.data(), ws.sel_rerun.p, rr.size() * 4, hipMemcpyDeviceToHost, c.stream));
  HIPCK(hipStreamSynchronize(c.stream));
  for (size_t i = 0; i < cand.size(); i += 2) pool.push_back(TopEntry{(int32_t)cand[i], cand[i + 1]});
  for (size_t i = 0; i < tie.size(); i += 2) pool.push_back(TopEntry{(int32_t)tie[i], tie[i + 1]});
  std::vector<uint32_t> rerun(db->generic);
  for (uint32_t slot : rr) rerun.push_back(db->slot_target[slot]);
  std::vector<int32_t> rs;
  if ((rc = rescore_locked(c, ws, db, qp, sc, rerun, rs))) return rc;
  for (size_t i = 0; i < rerun.size(); i++) pool.push_back(TopEntry{rs[i], rerun[i]});
  done = true;
  return SW_OK;
}

// the k best of `pool`, or of the full score vector when the device selection did not finish
int finish_top_locked(Context& c, Workspace& ws, sw_db* db, const QueryPrep& qp, Scoring sc, size_t k, bool device_done,
                      std::vector<TopEntry>& pool, std::vector<TopEntry>& top) {
  if (!device_done) {
    std::vector<int32_t> scores;
    int rc = db_scores_locked(c, ws, db, qp, sc, scores);
    if (rc) return rc;
    pool.resize(db->n);
    for (uint32_t i = 0; i < db->n; i++) pool[i] = TopEntry{scores[i], i};
  }
  const size_t kk = std::min(k, pool.size());
  std::partial_sort(pool.begin(), pool.begin() + kk, pool.end(), top_before);
  top.assign(pool.begin(), pool.begin() + kk);
  return SW_OK;
}

bool device_topk(const sw_db* db, const QueryPrep& qp, Scoring sc, size_t k) {
  return fast_path(db, qp, sc) && !g_host_topk && k <= 4096;
}

// The k best (score, target) of the database for one query.  Selection runs on the device (sw_topk.hip): only
// the candidates and the slots that need an exact re-run are downloaded.  Falls back to the full download when
// the tie list overflows (e.g. a database where every target scores the same) or with SW_HIP_TOPK=host.
int db_topk_locked(Context& c, Workspace& ws, sw_db* db, const QueryPrep& qp, Scoring sc, size_t k, std::vector<TopEntry>& top) {
  StreamDrain drain(c.stream);
  top.clear();
  k = std::min<size_t>(k, db->n);
  if (k == 0) return SW_OK;
  const uint32_t Q = (uint32_t)qp.d.sym.size();
  bool device_done = false;
  std::vector<TopEntry> pool;
  if (device_topk(db, qp, sc, k)) {
    int rc = score_pass_locked(c, ws, db, qp.qwords.data(), qp.qwords.size(), Q, sc, nullptr, false, nullptr, nullptr, nullptr,
                               &qp.qwords_small);
    if (rc) return rc;
    if ((rc = select_pool_locked(c, ws, db, qp, sc, k, (const int32_t*)ws.slot_scores.p, Q, pool, device_done))) return rc;
    if (!device_done) {
      // undo the accounting of the pass we are about to repeat through the host path
      pool.clear();
      c.stats.score_launches -= 1;
      c.stats.score_cells -= (uint64_t)Q * db->packed_residues;
    }
  }
  return finish_top_locked(c, ws, db, qp, sc, k, device_done, pool, top);
}

// Several queries in ONE score pass (sw_score.hip, MULTI): stacked into one column of row blocks, every query on a
// block boundary, one score array per query; then the same device selection per query.  What a short query alone
// amortises badly -- a strip's substitution tables, the unit bookkeeping, the start of the wave pipeline -- is paid
// once per batch.  qps[lo..hi) all pass device_topk() and agree on which targets take the full-matrix path
// (db_topk_phase decides that).
int db_topk_multi_locked(Context& c, Workspace& ws, sw_db* db, const std::vector<QueryPrep>& qps, size_t lo, size_t hi,
                         Scoring sc, size_t k, std::vector<std::vector<TopEntry>>& tops) {
  StreamDrain drain(c.stream);
  k = std::min<size_t>(k, db->n);
  std::vector<uint32_t> qwords, blkinfo;
  ScoreBatch batch{};
  for (size_t q = lo; q < hi; q++) {
    const uint32_t Q = (uint32_t)qps[q].d.sym.size();
    const uint32_t nb = (Q + kScoreB - 1) / kScoreB;
    for (uint32_t b = 0; b < nb; b++) {
      const uint32_t rows = std::min<uint32_t>(kScoreB, Q - b * kScoreB);
      blkinfo.push_back(rows | (b == 0 ? kBlkFirst : 0u) | (b + 1 == nb ? kBlkLast : 0u) | ((uint32_t)(q - lo) << kBlkQueryShift));
      qwords.push_back(qps[q].qwords[b]);
    }
    batch.rows += Q;
  }
  qwords.push_back(0);   // the kernel fetches both one block ahead
  blkinfo.push_back(0);
  batch.blkinfo = blkinfo.data();
  batch.n_blocks = (uint32_t)blkinfo.size() - 1;
  batch.n_queries = (uint32_t)(hi - lo);
  const uint32_t Q0 = (uint32_t)qps[lo].d.sym.size();
  int rc = score_pass_locked(c, ws, db, qwords.data(), qwords.size(), Q0, sc, nullptr, false, &batch);
  if (rc) return rc;
  const size_t n_slots = db->slot_target.size();
  std::vector<size_t> redo;
  for (size_t q = lo; q < hi; q++) {
    std::vector<TopEntry> pool;
    bool done = false;
    if ((rc = select_pool_locked(c, ws, db, qps[q], sc, k, (const int32_t*)ws.slot_scores.p + (q - lo) * n_slots,
                                 q == lo ? batch.rows : 0, pool, done)))
      return rc;
    if (!done) {
      redo.push_back(q);  // (after the loop: the host path runs a score pass of its own over the same buffers)
      continue;
    }
    if ((rc = finish_top_locked(c, ws, db, qps[q], sc, k, true, pool, tops[q]))) return rc;
  }
  for (size_t q : redo) {
    std::vector<TopEntry> pool;
    if ((rc = finish_top_locked(c, ws, db, qps[q], sc, k, false, pool, tops[q]))) return rc;
  }
  return SW_OK;
}

}  // namespace
bool swhip::prep_query(const char* query, QueryPrep& qp) {
  if (!decode(query, qp.d)) return false;
  const size_t Q = qp.d.sym.size();
  // one word per block of rows, 2 bits per row, plus a padding word; for both block sizes of the score kernel
  qp.qwords.assign((Q + kScoreB - 1) / kScoreB + 1, 0);
  qp.qwords_small.assign((Q + kScoreBSmall - 1) / kScoreBSmall + 1, 0);
  for (size_t i = 0; i < Q; i++) {
    int code = qp.d.cp(i) < 128 ? nt_code((unsigned char)qp.d.cp(i)) : -1;
    if (code < 0) {
      qp.acgt = false;
      code = 0;
    }
    qp.qwords[i / kScoreB] |= (uint32_t)code << (2 * (i % kScoreB));
    qp.qwords_small[i / kScoreBSmall] |= (uint32_t)code << (2 * (i % kScoreBSmall));
  }
  return true;
}

// ---- a score pass whose inter-unit boundary columns stay resident (the work-package fast path, sw_package.cpp) -------
swhip::PersistPass::~PersistPass() {
  for (hipEvent_t e : {ev_k0, ev_k1, ev_done})
    if (e) (void)hipEventDestroy(e);
}
// The pass is QUEUED (kernel, copies of its outputs into the pass's page-locked host buffers, a completion event) and
// not waited for: the caller may queue the next query's pass on another workspace behind it and do host work.  The
// stream must be drained before `out` goes away if db_score_persist_end_locked is not reached.
int swhip::db_score_persist_begin_locked(Context& c, Workspace& ws, sw_db* db, const QueryPrep& qp, Scoring sc, PersistPass& out) {
  if (!fast_path(db, qp, sc)) return fail(SW_E_DOMAIN, "internal: persist pass outside the packed kernel's domain");
  const uint32_t Q = (uint32_t)qp.d.sym.size();
  if (!out.ev_done) {
    HIPCK(hipEventCreate(&out.ev_k0));
    HIPCK(hipEventCreate(&out.ev_k1));
    HIPCK(hipEventCreateWithFlags(&out.ev_done, hipEventDisableTiming));
  }
  int block_rows = kScoreB;
  int rc = score_pass_locked(c, ws, db, qp.qwords.data(), qp.qwords.size(), Q, sc, nullptr, true, nullptr, out.ev_k0, out.ev_k1,
                             &qp.qwords_small, &block_rows);
  if (rc) return rc;
  const size_t slot_bytes = (db->slot_target.size() * 4 + 255) & ~(size_t)255, strip_bytes = (size_t)db->n_elems * 256;
  out.host.a.resize(slot_bytes + strip_bytes + 256);
  int32_t* const h_slot = (int32_t*)out.host.a.data();
  uint32_t* const h_strip = (uint32_t*)(out.host.a.data() + slot_bytes);
  uint32_t* const h_err = (uint32_t*)(out.host.a.data() + slot_bytes + strip_bytes);
  *h_err = 0;
  out.slot = h_slot;
  out.strip_best = h_strip;
  out.kerr = h_err;
  out.Q = Q;
  HIPCK(hipMemcpyAsync(h_slot, ws.slot_scores.p, db->slot_target.size() * 4, hipMemcpyDeviceToHost, c.stream));
  if (strip_bytes) HIPCK(hipMemcpyAsync(h_strip, ws.strip_best.p, strip_bytes, hipMemcpyDeviceToHost, c.stream));
  HIPCK(hipMemcpyAsync(h_err, (uint32_t*)ws.ticket.p + 32, 4, hipMemcpyDeviceToHost, c.stream));
  HIPCK(hipEventRecord(out.ev_done, c.stream));
  out.units_keep = db->h_units;
  out.units = out.units_keep.get();
  out.groups = &db->groups;
  out.slot_target = &db->slot_target;
  out.d_ring = (const uint32_t*)ws.gbound.p;
  out.ring_stride = (size_t)((Q + (uint32_t)block_rows - 1) / (uint32_t)block_rows) * (size_t)block_rows * 64;
  out.n_elems = db->n_elems;
  return SW_OK;
}
int swhip::db_score_persist_end_locked(Context& c, sw_db* db, PersistPass& out) {
  HIPCK(hipEventSynchronize(out.ev_done));
  return score_pass_finish_locked(c, db, out.Q, *out.kerr, out.ev_k0, out.ev_k1);
}
int swhip::db_score_persist_locked(Context& c, Workspace& ws, sw_db* db, const QueryPrep& qp, Scoring sc, PersistPass& out) {
  StreamDrain drain(c.stream);
  int rc = db_score_persist_begin_locked(c, ws, db, qp, sc, out);
  if (rc) return rc;
  return db_score_persist_end_locked(c, db, out);
}
size_t swhip::db_persist_ring_bytes(const sw_db* db, uint32_t Q, int cus) {
  const uint32_t W = (uint32_t)score_waves_per_group(db->n_elems, cus, (uint32_t)db->groups.size());
  const size_t n_units = units_bound(db->n_elems, db->groups.size(), W, (uint32_t)cus * (uint32_t)score_workgroups_per_cu((int)W));
  return n_units * (size_t)((Q + kScoreB - 1) / kScoreB) * kScoreB * 256;
}

// ---- the two phases of a search (used by the single-database entry points below and by sw_node.cpp) -------------
Context* swhip::db_context(sw_db* db) { return db ? db->ctx : nullptr; }
const sw_kernel_stats& swhip::db_last_stats(const sw_db* db) { return db->last_stats; }

int swhip::db_topk_phase(sw_db* db, const std::vector<QueryPrep>& qps, Scoring sc, size_t k,
                         std::vector<std::vector<TopEntry>>& tops) {
  tops.assign(qps.size(), {});
  CtxLock L(db->ctx);
  if (L.rc) return L.rc;
  reset_stats(*L.c);
  // Consecutive short queries share one score pass (db_topk_multi_locked).  They must agree on which targets take
  // the full-matrix path (T < 2Q decides whether a target whose length is a multiple of 64 loses its last column,
  // and that is built into a strip's tables): no such length of this database may lie in [2 Qmin, 2 Qmax).
  // SW_HIP_QUERY_BATCH=0 switches the batching off, =n caps a batch at n queries (default 16).
  static const size_t max_batch = [] {
    const char* e = getenv("SW_HIP_QUERY_BATCH");
    return e ? (size_t)std::max(0, atoi(e)) : (size_t)16;
  }();
  constexpr uint32_t kBatchQueryRows = 1024, kBatchRows = 4096;
  const size_t n_slots = db->slot_target.size();
  const size_t k_eff = std::min<size_t>(k, db->n);
  size_t q = 0;
  while (q < qps.size()) {
    size_t hi = q + 1;
    auto eligible = [&](const QueryPrep& qp) {
      return k_eff > 0 && device_topk(db, qp, sc, k_eff) && qp.d.sym.size() <= kBatchQueryRows;
    };
    if (max_batch >= 2 && eligible(qps[q])) {
      uint32_t qmin = (uint32_t)qps[q].d.sym.size(), qmax = qmin;
      uint64_t rows = ((uint64_t)qmin + kScoreB - 1) / kScoreB * kScoreB;
      while (hi < qps.size() && hi - q < max_batch && eligible(qps[hi])) {
        const uint32_t Q = (uint32_t)qps[hi].d.sym.size();
        const uint32_t lo2 = std::min(qmin, Q), hi2 = std::max(qmax, Q);
        const uint64_t r2 = rows + ((uint64_t)Q + kScoreB - 1) / kScoreB * kScoreB;
        const auto it = std::lower_bound(db->len64.begin(), db->len64.end(), 2 * lo2);
        if (r2 > kBatchRows || (hi - q + 1) * n_slots * 4 > ((size_t)1 << 30) || (it != db->len64.end() && *it < 2 * (uint64_t)hi2)) break;
        qmin = lo2, qmax = hi2, rows = r2;
        hi++;
      }
    }
    int rc;
    if (hi - q >= 2) rc = db_topk_multi_locked(*L.c, L.c->ws, db, qps, q, hi, sc, k, tops);
    else rc = db_topk_locked(*L.c, L.c->ws, db, qps[q], sc, k, tops[q]);
    if (rc) return rc;
    q = hi;
  }
  db->last_stats = L.c->stats;
  return SW_OK;
}

int swhip::db_align_phase(sw_db* db, const std::vector<QueryPrep>& qps, Scoring sc, const std::vector<HitRef>& refs,
                          sw_hit* hits_out) {
  const size_t total = refs.size();
  for (size_t i = 0; i < total; i++) hits_out[i] = sw_hit{};
  if (total == 0) return SW_OK;
  std::vector<PairSpec> specs;
  std::vector<Decoded> tdec(total);
  std::vector<PairOut> po(total);
  ArenaLease lease;
  OpsArena& arena = lease.a;
  specs.reserve(total);
  for (size_t i = 0; i < total; i++) {
    const HitRef& r = refs[i];
    if (r.query >= qps.size() || r.target >= db->n) return fail(SW_E_ARG, "internal: hit reference out of range");
    const QueryPrep& qp = qps[r.query];
    decode(db->flat.data() + db->off[r.target], (size_t)(db->off[r.target + 1] - db->off[r.target]), tdec[i]);
    specs.push_back(PairSpec{qp.d.sym.data(), (uint32_t)qp.d.sym.size(), tdec[i].sym.data(), (uint32_t)tdec[i].sym.size(),
                             db_pair_path((uint32_t)qp.d.sym.size(), (uint32_t)tdec[i].sym.size(), sc)});
  }
  {
    CtxLock L(db->ctx);
    if (L.rc) return L.rc;
    const double ms0 = L.c->stats.exact_kernel_ms;
    const uint64_t cells0 = L.c->stats.exact_cells;
    const int rc = run_exact(*L.c, L.c->ws, specs.data(), total, sc, true, po.data(), &arena);
    if (rc) return rc;
    db->last_stats.exact_kernel_ms += L.c->stats.exact_kernel_ms - ms0;
    db->last_stats.exact_cells += L.c->stats.exact_cells - cells0;
  }
  for (size_t i = 0; i < total; i++) {
    const HitRef& r = refs[i];
    char *qa = nullptr, *ta = nullptr;
    size_t qlen = 0;
    const bool built = build_alignment(qps[r.query].d, tdec[i], po[i], arena.data() + po[i].ops_off, &qa, &ta, &qlen);
    if (!built || po[i].score != r.score) {
      str_free(qa);
      str_free(ta);
      sw_hits_free(hits_out, (int)i);
      if (!built) return fail(SW_E_NOMEM, "out of memory");
      return fail(SW_E_HIP, "internal: score kernel and exact kernel disagree on target " + std::to_string(r.target) +
                                " (" + std::to_string(r.score) + " vs " + std::to_string(po[i].score) + ")");
    }
    sw_hit& h = hits_out[i];
    h.target_index = r.target;
    h.score = po[i].score;
    h.length = (uint32_t)qlen;
    h.max_x = (uint64_t)po[i].bx - 1;
    h.max_y = (uint64_t)po[i].bj - 1;
    h.query_aln = qa;
    h.target_aln = ta;
  }
  return SW_OK;
}

// Builds a database on a given context from n sequences (sw_db_create / sw_node_db_create).
sw_db* swhip::db_create_on(Context* ctx, const char* flat, const uint64_t* offsets, uint64_t n) {
  if ((!flat && n) || !offsets) {
    fail(SW_E_ARG, "null argument");
    return nullptr;
  }
  for (uint64_t i = 0; i < n; i++)
    if (offsets[i + 1] < offsets[i]) {
      fail(SW_E_ARG, "offsets must be non-decreasing");
      return nullptr;
    }
  std::vector<uint64_t> off(n + 1);
  for (uint64_t i = 0; i <= n; i++) off[i] = offsets[i] - offsets[0];
  std::string f(n ? flat + offsets[0] : "", n ? flat + offsets[n] : "");
  CtxLock L(ctx);
  if (L.rc) return nullptr;
  return db_build(L.c, std::move(f), std::move(off));
}
// ... from sequences already gathered by the caller (the node sharder): takes ownership, no second copy
// Takes the host copy of the sequences out of a database that is about to be freed (the temporary databases of the
// work-package fast path hand their buffer on to the next one).
std::string swhip::db_take_flat(sw_db* db) {
  std::string out;
  if (db) out.swap(db->flat);
  return out;
}
sw_db* swhip::db_create_owned(Context* ctx, std::string&& flat, std::vector<uint64_t>&& off, bool known_acgt) {
  CtxLock L(ctx);
  if (L.rc) return nullptr;
  return db_build(L.c, std::move(flat), std::move(off), known_acgt);
}
sw_db* swhip::db_create_temp_locked(Context& c, std::string&& flat, std::vector<uint64_t>&& off) {
  return db_build(&c, std::move(flat), std::move(off), true, true);
}
void swhip::db_release_locked(sw_db* db) { db_release(db); }
sw_db* swhip::db_open_on(Context* ctx, const char* path) {
  if (!path) {
    fail(SW_E_ARG, "null argument");
    return nullptr;
  }
  CtxLock L(ctx);
  if (L.rc) return nullptr;
  return db_open_locked(L.c, path);
}
const std::string& swhip::db_flat(const sw_db* db) { return db->flat; }
const std::vector<uint64_t>& swhip::db_offsets(const sw_db* db) { return db->off; }

// ---- the allocation-free variant of sw_db_scores --------------------------------------------------------------------
namespace {
// One pair through the exact kernel, score only, with every buffer taken from the fixed workspace / host arena.
int exact_one_fixed(Context& c, HostArena& arena, const uint16_t* qsym, uint32_t Q, const unsigned char* tbytes, uint32_t T,
                    int path, Scoring sc, int32_t* score_out) {
  Workspace& ws = c.na;
  const size_t mark = arena.used;
  struct Restore {
    HostArena& a;
    size_t m;
    ~Restore() { a.used = m; }
  } restore{arena, mark};
  const int rpl = 4;
  const PathParams pp = resolve_path(path, Q, T);
  const uint32_t maxcol = std::max(pp.Tlim, pp.T0);
  const uint32_t n_units = std::max<uint32_t>(exact_nstrips(Q, rpl), 1);
  auto align16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
  const size_t in_units = align16(sizeof(ExactPair)), in_order = in_units + align16((size_t)n_units * 4),
               in_q = in_order + align16((size_t)n_units * 4), in_t = in_q + align16((size_t)Q * 2 + 2),
               in_bytes = in_t + align16((size_t)T * 2 + 16);
  const size_t ctl_words = 64 + 1, z_bnd = (ctl_words * 4 + 63) & ~(size_t)63,
               z_bytes = z_bnd + align16(std::max<size_t>((size_t)n_units * exact_bnd_pitch(maxcol) * 4, 16));
  int rc;
  if ((rc = ws.xin.ensure(in_bytes))) return rc;
  if ((rc = ws.xzero.ensure(z_bytes))) return rc;
  if ((rc = ws.xout.ensure(sizeof(ExactOut)))) return rc;
  if ((rc = ws.xrecs.ensure((size_t)n_units * sizeof(ExactRec)))) return rc;
  unsigned char* h = (unsigned char*)arena.take(in_bytes);
  ExactOut* ho = (ExactOut*)arena.take(sizeof(ExactOut));
  uint32_t* herr = (uint32_t*)arena.take(16);
  if (!h || !ho || !herr) return fail(SW_E_LIMIT, "sw_db_scores_noalloc: a target that needs the int32 re-run is too long for the host arena");
  unsigned char* const d_in = (unsigned char*)ws.xin.p;
  unsigned char* const d_zero = (unsigned char*)ws.xzero.p;
  ExactPair e{};
  e.q = (const uint16_t*)(d_in + in_q);
  e.t = (const uint16_t*)(d_in + in_t);
  e.Q = Q;
  e.T = T;
  e.Tlim = pp.Tlim;
  e.T0 = pp.T0;
  e.phaseQp = pp.phaseQp;
  e.straight = pp.straight;
  e.bnd = (int32_t*)(d_zero + z_bnd);
  e.bnd_pitch = exact_bnd_pitch(maxcol);
  e.dirs = nullptr;
  e.nkb = exact_nsteps(maxcol, Q, rpl) / 16;
  e.ops = nullptr;
  e.first_unit = 0;
  e.n_units = n_units;
  memcpy(h, &e, sizeof(e));
  uint32_t* up = (uint32_t*)(h + in_units);
  uint32_t* uo = (uint32_t*)(h + in_order);
  for (uint32_t u = 0; u < n_units; u++) up[u] = 0, uo[u] = u;
  memcpy(h + in_q, qsym, (size_t)Q * 2);
  uint16_t* ts = (uint16_t*)(h + in_t);
  for (uint32_t i = 0; i < T; i++) ts[i] = tbytes[i];
  HIPCK(hipMemcpyAsync(d_in, h, in_bytes, hipMemcpyHostToDevice, c.stream));
  HIPCK(hipMemsetAsync(d_zero, 0, z_bytes, c.stream));
  ExactWork w{};
  w.pairs = (const ExactPair*)d_in;
  w.outs = (ExactOut*)ws.xout.p;
  w.unit_pair = (const uint32_t*)(d_in + in_units);
  w.unit_order = (const uint32_t*)(d_in + in_order);
  w.n_units = n_units;
  w.n_pairs = 1;
  w.rpl = rpl;
  w.ticket = (uint32_t*)d_zero;
  w.err = (uint32_t*)d_zero + 32;
  w.pair_done = (uint32_t*)d_zero + 64;
  w.recs = (ExactRec*)ws.xrecs.p;
  w.wait_ticks = exact_wait_ticks();
  HIPCK(launch_exact(w, sc, c.stream));
  HIPCK(hipMemcpyAsync(ho, ws.xout.p, sizeof(ExactOut), hipMemcpyDeviceToHost, c.stream));
  HIPCK(hipMemcpyAsync(herr, w.err, 4, hipMemcpyDeviceToHost, c.stream));
  HIPCK(hipStreamSynchronize(c.stream));
  // (a wave whose hand-over wait ran out has ended without a result: the error word says so)
  if (*herr != 0 || ho->nops == 0xffffffffu) return fail(SW_E_HIP, "exact kernel: strip pipeline wait timed out");
  *score_out = ho->score;
  c.stats.exact_cells += (uint64_t)Q * T;
  c.stats.rerun_pairs += 1;
  return SW_OK;
}

int db_scores_noalloc(sw_db* db, const char* query, AlignmentScores as, int32_t* scores_out) {
  if (!db || !query || !scores_out) return fail(SW_E_ARG, "null argument");
  const Scoring sc = to_scoring(as);
  if (!fast_domain(sc))
    return fail(SW_E_DOMAIN, "sw_db_scores_noalloc: scoring outside match > 0, mismatch < 0, gap < 0; use sw_db_scores");
  for (uint32_t ti : db->generic)
    if (db->off[ti + 1] > db->off[ti])
      return fail(SW_E_DOMAIN, "sw_db_scores_noalloc: the database holds a sequence with symbols outside ACGT (index " +
                                   std::to_string(ti) + "); use sw_db_scores");
  const size_t Qlen = strlen(query);
  for (size_t i = 0; i < Qlen; i++)
    if (nt_code((unsigned char)query[i]) < 0)
      return fail(SW_E_DOMAIN, "sw_db_scores_noalloc: the query holds symbols outside ACGT; use sw_db_scores");
  if (Qlen > 0x10000) return fail(SW_E_LIMIT, "sw_db_scores_noalloc: query longer than 65536 residues");
  for (uint64_t i = 0; i < db->n; i++) scores_out[i] = 0;
  if (Qlen == 0 || db->groups.empty()) return SW_OK;
  const uint32_t Q = (uint32_t)Qlen;
  CtxLock L(db->ctx);
  if (L.rc) return L.rc;
  Context& c = *L.c;
  HostArena& arena = c.na_host;
  arena.reset();
  reset_stats(c);
  const size_t n_qwords = (Q + kScoreB - 1) / kScoreB + 1;
  uint32_t* qwords = (uint32_t*)arena.take(n_qwords * 4);
  uint16_t* qsym = (uint16_t*)arena.take((size_t)Q * 2);
  const size_t n_slots = db->slot_target.size();
  int32_t* slot = (int32_t*)arena.take(n_slots * 4);
  if (!qwords || !qsym || !slot)
    return fail(SW_E_LIMIT, "sw_db_scores_noalloc: the database is too large for the pre-reserved host arena (" +
                                std::to_string(n_slots) + " slots)");
  memset(qwords, 0, n_qwords * 4);
  for (uint32_t i = 0; i < Q; i++) {
    qwords[i / kScoreB] |= (uint32_t)nt_code((unsigned char)query[i]) << (2 * (i % kScoreB));
    qsym[i] = (unsigned char)query[i];
  }
  StreamDrain drain(c.stream);
  int rc = score_pass_locked(c, c.na, db, qwords, n_qwords, Q, sc, &arena);
  if (rc) return rc;
  uint32_t* kerr = (uint32_t*)arena.take(4);
  if (!kerr) return fail(SW_E_LIMIT, "sw_db_scores_noalloc: host arena exhausted");
  *kerr = 0;
  HIPCK(hipMemcpyAsync(slot, c.na.slot_scores.p, n_slots * 4, hipMemcpyDeviceToHost, c.stream));
  HIPCK(hipMemcpyAsync(kerr, (uint32_t*)c.na.ticket.p + 32, 4, hipMemcpyDeviceToHost, c.stream));
  HIPCK(hipStreamSynchronize(c.stream));
  if ((rc = score_pass_finish_locked(c, db, Q, *kerr))) return rc;
  const uint32_t Qp = roundup64(Q);
  for (size_t k = 0; k < n_slots; k++) {
    const uint32_t ti = db->slot_target[k];
    if (ti == UINT32_MAX) continue;
    const uint32_t T = (uint32_t)(db->off[ti + 1] - db->off[ti]);
    if (!needs_exact(T, slot[k], Q, Qp, sc)) {
      scores_out[ti] = slot[k];
      continue;
    }
    // saturated / quirk lengths: the int32 exact kernel, one pair per submission, same fixed scratch
    int32_t v = 0;
    rc = exact_one_fixed(c, arena, qsym, Q, (const unsigned char*)db->flat.data() + db->off[ti], T, db_pair_path(Q, T, sc), sc, &v);
    if (rc) return rc;
    scores_out[ti] = v;
  }
  return SW_OK;
}
}  // namespace

extern "C" {

int sw_db_scores_noalloc(sw_db* db, const char* query, AlignmentScores as, int32_t* scores_out) {
  return swhip::guarded<int>(SW_E_NOMEM, [&] { return db_scores_noalloc(db, query, as, scores_out); });
}
uint64_t sw_hip_noalloc_scratch_bytes(void) { return noalloc_scratch_bytes(); }

sw_db* sw_db_create(const char* flat, const uint64_t* offsets, uint64_t n) {
  return swhip::guarded<sw_db*>(nullptr, [&]() -> sw_db* {
    Context* c = nullptr;
    if (ctx_get(-1, &c)) return nullptr;
    return db_create_on(c, flat, offsets, n);
  });
}

sw_db* sw_db_load_fasta(const char* path) {
  // cli/__main__.py:19-34 through the one parser of sw_worker.cpp
  sw_fasta* f = sw_fasta_parse(path);
  if (!f) return nullptr;
  if (sw_fasta_count(f) == 0) {
    sw_fasta_free(f);
    fail(SW_E_ARG, "no FASTA records");
    return nullptr;
  }
  sw_db* db = sw_db_from_fasta(f);
  sw_fasta_free(f);
  return db;
}

int sw_db_save(sw_db* db, const char* path) {
  return swhip::guarded<int>(SW_E_NOMEM, [&]() -> int {
    if (!db || !path) return fail(SW_E_ARG, "null argument");
    CtxLock L(db->ctx);
    if (L.rc) return L.rc;
    return db_save_locked(db, path);
  });
}
sw_db* sw_db_open(const char* path) {
  return swhip::guarded<sw_db*>(nullptr, [&]() -> sw_db* {
    Context* c = nullptr;
    if (ctx_get(-1, &c)) return nullptr;
    return db_open_on(c, path);
  });
}

void sw_db_free(sw_db* db) {
  if (!db) return;
  CtxLock L(db->ctx);  // (a failed hipSetDevice still frees the host side)
  db_release(db);
}
uint64_t sw_db_size(const sw_db* db) { return db ? db->n : 0; }
uint64_t sw_db_residues(const sw_db* db) { return db ? db->residues : 0; }

int sw_db_scores(sw_db* db, const char* query, AlignmentScores as, int32_t* scores_out) {
  return swhip::guarded<int>(SW_E_NOMEM, [&]() -> int {
  if (!db || !query || !scores_out) return fail(SW_E_ARG, "null argument");
  QueryPrep qp;
  if (!prep_query(query, qp)) return fail(SW_E_ARG, "query is not valid UTF-8");
  const Scoring sc = to_scoring(as);
  std::vector<int32_t> scores;
  CtxLock L(db->ctx);
  if (L.rc) return L.rc;
  reset_stats(*L.c);
  const int rc = db_scores_locked(*L.c, L.c->ws, db, qp, sc, scores);
  if (rc) return rc;
  memcpy(scores_out, scores.data(), scores.size() * sizeof(int32_t));
  return SW_OK;
  });
}

static int db_search_topk_batch(sw_db* db, const char* const* queries, size_t n_queries, AlignmentScores as, int k,
                                sw_hit* hits, int* n_hits) {
  if (!db || (!queries && n_queries) || !hits || !n_hits || k < 0) return fail(SW_E_ARG, "bad argument");
  for (size_t q = 0; q < n_queries; q++) n_hits[q] = 0;
  if (n_queries == 0 || k == 0) return SW_OK;
  std::vector<QueryPrep> qps(n_queries);
  for (size_t q = 0; q < n_queries; q++)
    if (!queries[q] || !prep_query(queries[q], qps[q]))
      return fail(SW_E_ARG, "query " + std::to_string(q) + " is not valid UTF-8");
  const Scoring sc = to_scoring(as);
  // score pass + selection per query, then ONE exact submission (end cell + traceback) for all reported hits
  std::vector<std::vector<TopEntry>> tops;
  int rc = db_topk_phase(db, qps, sc, (size_t)k, tops);
  if (rc) return rc;
  std::vector<HitRef> refs;
  for (size_t q = 0; q < n_queries; q++)
    for (const TopEntry& e : tops[q]) refs.push_back(HitRef{(uint32_t)q, e.target, e.score});
  std::vector<sw_hit> flat_hits(refs.size());
  rc = db_align_phase(db, qps, sc, refs, flat_hits.data());
  if (rc) return rc;
  size_t at = 0;
  for (size_t q = 0; q < n_queries; q++) {
    for (size_t i = 0; i < tops[q].size(); i++) hits[q * (size_t)k + i] = flat_hits[at++];
    n_hits[q] = (int)tops[q].size();
  }
  return SW_OK;
}

int sw_db_search_topk_batch(sw_db* db, const char* const* queries, size_t n_queries, AlignmentScores as, int k,
                            sw_hit* hits, int* n_hits) {
  return swhip::guarded<int>(SW_E_NOMEM, [&] { return db_search_topk_batch(db, queries, n_queries, as, k, hits, n_hits); });
}

int sw_db_search_topk(sw_db* db, const char* query, AlignmentScores as, int k, sw_hit* hits, int* n_hits) {
  if (!query) return fail(SW_E_ARG, "bad argument");
  return sw_db_search_topk_batch(db, &query, 1, as, k, hits, n_hits);
}

void sw_hits_free(sw_hit* hits, int n_hits) {
  if (!hits) return;
  for (int i = 0; i < n_hits; i++) {
    str_free(hits[i].query_aln);
    str_free(hits[i].target_aln);
    hits[i].query_aln = hits[i].target_aln = nullptr;
  }
}

}  // extern "C"

// The worker's capacity benchmark on one device: internal/worker/worker_rust_benchmark.go:38-68 measures one CPU
// thread on one synthetic pair; a GPU worker runs work packages through the batch path, so that is what is timed
// here: 1 query (4096 nt) x 16384 targets (1024 nt) of seeded uniform DNA, match 1 / mismatch -1 / gap -2 (the
// reference's benchmark scores, worker_rust_benchmark.go:10-18).  Cells per second, 0 on failure.
double swhip::benchmark_cups_on(Context* ctx) {
  const uint32_t nQ = 4096, nT = 1024, n = 16384;
  uint64_t st = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() {
    st += 0x9E3779B97F4A7C15ull;
    uint64_t z = st;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  };
  const char nt[4] = {'A', 'C', 'G', 'T'};
  std::string flat((size_t)n * nT, 'A'), query(nQ, 'A');
  for (auto& ch : flat) ch = nt[rnd() & 3];
  for (auto& ch : query) ch = nt[rnd() & 3];
  std::vector<uint64_t> off(n + 1);
  for (uint32_t i = 0; i <= n; i++) off[i] = (uint64_t)i * nT;
  sw_db* db = db_create_on(ctx, flat.data(), off.data(), n);
  if (!db) return 0.0;
  std::vector<int32_t> sc(n);
  AlignmentScores as{(uint16_t)-2, 1, (uint16_t)-1};
  double best = 0.0;
  for (int rep = 0; rep < 3; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    if (sw_db_scores(db, query.c_str(), as, sc.data()) != SW_OK) {
      sw_db_free(db);
      return 0.0;
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    best = std::max(best, (double)nQ * nT * n / secs);
  }
  sw_db_free(db);
  return best;
}

extern "C" float sw_hip_benchmark_cups(void) {
  // Returned as an integer-valued float32 because the master's field is a pydantic int
  // (master/api_models/worker.py:7-16).
  return swhip::guarded<float>(0.0f, [&]() -> float {
    Context* c = nullptr;
    if (ctx_get(-1, &c)) return 0.0f;
    return (float)(double)(uint64_t)benchmark_cups_on(c);
  });
}
