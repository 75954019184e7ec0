// tfrecord.hip -- host-side (no device code) native input reader: TFRecord framing + tf.train.Example /
// tf.train.SequenceExample wire-format decode for the two YouTube-8M reader contracts:
//   * frame level  (W/readers.py:189-259  YT8MFrameFeatureReader.prepare_reader):  context {"video_id": bytes,
//     "labels": int64 list}, feature_lists {<name>: one bytes feature per frame, feature_size uint8 each}
//     -> raw uint8 [max_frames, sum(sizes)] (NOT dequantised: the device kernel does that), num_frames = min(n, max_frames)
//   * video level  (W/readers.py:94-125   YT8MAggregatedFeatureReader.prepare_reader): features {"video_id", "labels",
//     <name>: float list of feature_size} -> float32 [sum(sizes)]
// Labels become a uint8 multi-hot of num_classes (sparse_to_dense / sparse_to_indicator: duplicates and order
// irrelevant, W/readers.py:120,217-220).  TFRecord framing: u64 length, u32 masked crc32c(length), payload, u32 masked
// crc32c(payload); mask(c) = ((c >> 15) | (c << 17)) + 0xa282ead8.  Everything here is byte / integer work: bit-exact.
#include <stdlib.h>
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>
#include "common.h"

namespace {

uint32_t g_crc_table[8][256];
std::once_flag g_crc_once;

void crc_build() {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);   // Castagnoli, reflected
    g_crc_table[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t) g_crc_table[t][i] = (g_crc_table[t - 1][i] >> 8) ^ g_crc_table[0][g_crc_table[t - 1][i] & 0xff];
}

void crc_init() { std::call_once(g_crc_once, crc_build); }   // the prefetch workers call this concurrently

uint32_t crc32c(const uint8_t* p, size_t n) {
  crc_init();
  uint32_t c = 0xffffffffu;
  while (n >= 8) {  // slicing-by-8
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = g_crc_table[7][lo & 0xff] ^ g_crc_table[6][(lo >> 8) & 0xff] ^ g_crc_table[5][(lo >> 16) & 0xff] ^ g_crc_table[4][lo >> 24] ^
        g_crc_table[3][hi & 0xff] ^ g_crc_table[2][(hi >> 8) & 0xff] ^ g_crc_table[1][(hi >> 16) & 0xff] ^ g_crc_table[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = (c >> 8) ^ g_crc_table[0][(c ^ *p++) & 0xff];
  return c ^ 0xffffffffu;
}

inline uint32_t mask_crc(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xa282ead8u; }

// ---- protobuf wire format ---------------------------------------------------------------------------------------
struct Span {
  const uint8_t* p;
  const uint8_t* e;
  bool ok;
};

bool varint(Span& s, uint64_t& v) {
  v = 0;
  for (int shift = 0; shift < 64 && s.p < s.e; shift += 7) {
    const uint8_t b = *s.p++;
    v |= (uint64_t)(b & 0x7f) << shift;
    if (!(b & 0x80)) return true;
  }
  s.ok = false;
  return false;
}

// reads one field header; for length-delimited fields returns the sub-span; skips unknown wire types
bool next_field(Span& s, uint32_t& field, uint32_t& wt, Span& sub, uint64_t& val) {
  if (s.p >= s.e) return false;
  uint64_t key;
  if (!varint(s, key)) return false;
  field = (uint32_t)(key >> 3);
  wt = (uint32_t)(key & 7);
  sub = Span{nullptr, nullptr, true};
  val = 0;
  switch (wt) {
    case 0: return varint(s, val);
    case 1: if (s.e - s.p < 8) { s.ok = false; return false; } memcpy(&val, s.p, 8); s.p += 8; return true;
    case 5: { if (s.e - s.p < 4) { s.ok = false; return false; } uint32_t t; memcpy(&t, s.p, 4); val = t; s.p += 4; return true; }
    case 2: {
      uint64_t len;
      if (!varint(s, len) || (uint64_t)(s.e - s.p) < len) { s.ok = false; return false; }
      sub = Span{s.p, s.p + len, true};
      s.p += len;
      return true;
    }
    default: s.ok = false; return false;
  }
}

// map<string, X> entry: key = 1 (string), value = 2 (message)
bool map_entry(Span entry, std::string& key, Span& value) {
  uint32_t f, wt; Span sub; uint64_t v;
  key.clear();
  value = Span{nullptr, nullptr, true};
  while (next_field(entry, f, wt, sub, v)) {
    if (f == 1 && wt == 2) key.assign((const char*)sub.p, sub.e - sub.p);
    else if (f == 2 && wt == 2) value = sub;
  }
  return entry.ok;
}

struct Reader {
  FILE* fh = nullptr;
  std::vector<uint8_t> buf;
  bool check_crc = true;
  int64_t records = 0;
};

// returns 1 record read, 0 clean EOF, <0 error
int read_record(Reader* r) {
  uint8_t hdr[12];
  const size_t got = fread(hdr, 1, 12, r->fh);
  if (got == 0) return 0;
  if (got != 12) return yt8m::fail(YT8M_E_BADARG, "tfrecord: truncated record header%s", "");
  uint64_t len;
  uint32_t lcrc;
  memcpy(&len, hdr, 8);
  memcpy(&lcrc, hdr + 8, 4);
  if (r->check_crc && mask_crc(crc32c(hdr, 8)) != lcrc) return yt8m::fail(YT8M_E_BADARG, "tfrecord: corrupt length crc%s", "");
  if (len > (1ull << 31)) return yt8m::fail(YT8M_E_SHAPE, "tfrecord: record larger than 2 GiB%s", "");
  r->buf.resize(len);
  if (len && fread(r->buf.data(), 1, len, r->fh) != len) return yt8m::fail(YT8M_E_BADARG, "tfrecord: truncated payload%s", "");
  uint32_t dcrc;
  if (fread(&dcrc, 1, 4, r->fh) != 4) return yt8m::fail(YT8M_E_BADARG, "tfrecord: truncated payload crc%s", "");
  if (r->check_crc && mask_crc(crc32c(r->buf.data(), len)) != dcrc) return yt8m::fail(YT8M_E_BADARG, "tfrecord: corrupt payload crc%s", "");
  r->records++;
  return 1;
}

// Feature message -> (kind, payload span): kind 1 bytes_list, 2 float_list, 3 int64_list
void parse_labels(Span feature, uint8_t* labels, int64_t num_classes) {
  uint32_t f, wt; Span sub; uint64_t v;
  while (next_field(feature, f, wt, sub, v)) {
    if (f != 3 || wt != 2) continue;                     // Int64List
    Span lst = sub;
    uint32_t f2, wt2; Span sub2; uint64_t v2;
    while (next_field(lst, f2, wt2, sub2, v2)) {
      if (f2 != 1) continue;
      if (wt2 == 0) { if ((int64_t)v2 >= 0 && (int64_t)v2 < num_classes) labels[v2] = 1; }
      else if (wt2 == 2) {                               // packed
        Span pk = sub2;
        uint64_t x;
        while (pk.p < pk.e && varint(pk, x)) if ((int64_t)x >= 0 && (int64_t)x < num_classes) labels[x] = 1;
      }
    }
  }
}

bool first_bytes(Span feature, Span& out) {             // Feature{bytes_list{value[0]}}
  uint32_t f, wt; Span sub; uint64_t v;
  while (next_field(feature, f, wt, sub, v)) {
    if (f != 1 || wt != 2) continue;
    Span lst = sub;
    uint32_t f2, wt2; Span sub2; uint64_t v2;
    while (next_field(lst, f2, wt2, sub2, v2))
      if (f2 == 1 && wt2 == 2) { out = sub2; return true; }
  }
  return false;
}

int64_t parse_floats(Span feature, float* dst, int64_t cap) {   // Feature{float_list{value...}}; returns count (or -1)
  uint32_t f, wt; Span sub; uint64_t v;
  int64_t n = 0;
  while (next_field(feature, f, wt, sub, v)) {
    if (f != 2 || wt != 2) continue;
    Span lst = sub;
    uint32_t f2, wt2; Span sub2; uint64_t v2;
    while (next_field(lst, f2, wt2, sub2, v2)) {
      if (f2 != 1) continue;
      if (wt2 == 5) { if (n < cap) { uint32_t t = (uint32_t)v2; memcpy(dst + n, &t, 4); } ++n; }
      else if (wt2 == 2) {
        const int64_t cnt = (sub2.e - sub2.p) / 4;
        for (int64_t i = 0; i < cnt; ++i) { if (n < cap) memcpy(dst + n, sub2.p + 4 * i, 4); ++n; }
      }
    }
  }
  return n;
}

void copy_id(Span s, char* dst, int64_t stride) {
  if (!dst || stride <= 0) return;
  memset(dst, 0, stride);
  if (s.p) memcpy(dst, s.p, (size_t)((s.e - s.p) < stride - 1 ? (s.e - s.p) : stride - 1));
}

}  // namespace

extern "C" uint32_t yt8m_crc32c(const void* data, int64_t n) { return crc32c(static_cast<const uint8_t*>(data), (size_t)n); }
extern "C" uint32_t yt8m_crc32c_masked(const void* data, int64_t n) { return mask_crc(crc32c(static_cast<const uint8_t*>(data), (size_t)n)); }

extern "C" int yt8m_tfrecord_open(const char* path, int check_crc, void** reader_out) {
  using namespace yt8m;
  YT8M_REQUIRE(path && reader_out, YT8M_E_BADARG, "null argument");
  FILE* fh = fopen(path, "rb");
  if (!fh) return fail(YT8M_E_BADARG, "tfrecord: cannot open %s", path);    // the reference raises IOError (W/train.py:193-195)
  Reader* r = new Reader();
  r->fh = fh;
  r->check_crc = check_crc != 0;
  *reader_out = r;
  return YT8M_OK;
}

extern "C" int yt8m_tfrecord_close(void* reader) {
  Reader* r = static_cast<Reader*>(reader);
  if (r) { if (r->fh) fclose(r->fh); delete r; }
  return YT8M_OK;
}

// Frame-level batch.  feature_names: nfeat C strings; feature_sizes: bytes per frame of each.  Outputs (host):
//   q [max_records, max_frames, D] uint8 (D = sum sizes; rows >= num_frames are ZERO bytes -- the device transform
//   zeroes them after dequantisation from num_frames, readers.py:186), num_frames [max_records] int32,
//   labels [max_records, num_classes] uint8, video_ids [max_records, id_stride] NUL-padded (may be NULL).
extern "C" int yt8m_tfrecord_read_frame_batch(void* reader, const char* const* feature_names, const int32_t* feature_sizes,
                                              int nfeat, int64_t max_frames, int64_t num_classes, int64_t max_records,
                                              uint8_t* q, int32_t* num_frames, uint8_t* labels, char* video_ids,
                                              int64_t id_stride, int64_t* n_read) {
  using namespace yt8m;
  YT8M_REQUIRE(reader && feature_names && feature_sizes && q && num_frames && labels && n_read, YT8M_E_BADARG, "null argument");
  YT8M_REQUIRE(nfeat >= 1 && max_frames >= 1 && num_classes >= 1 && max_records >= 0, YT8M_E_SHAPE, "bad sizes");
  Reader* r = static_cast<Reader*>(reader);
  int64_t D = 0;
  std::vector<int64_t> off(nfeat);
  for (int i = 0; i < nfeat; ++i) { off[i] = D; D += feature_sizes[i]; }
  *n_read = 0;
  for (int64_t rec = 0; rec < max_records; ++rec) {
    const int rc = read_record(r);
    if (rc == 0) break;
    if (rc < 0) return rc;
    uint8_t* qrow = q + rec * max_frames * D;
    uint8_t* lrow = labels + rec * num_classes;
    memset(qrow, 0, (size_t)(max_frames * D));
    memset(lrow, 0, (size_t)num_classes);
    Span ex{r->buf.data(), r->buf.data() + r->buf.size(), true};
    uint32_t f, wt; Span sub; uint64_t v;
    int64_t nf = -1;
    std::vector<char> seen(nfeat, 0);
    Span vid{nullptr, nullptr, true};
    while (next_field(ex, f, wt, sub, v)) {
      if (wt != 2) continue;
      if (f == 1) {                                      // context: Features
        Span feats = sub;
        uint32_t f2, wt2; Span sub2; uint64_t v2;
        while (next_field(feats, f2, wt2, sub2, v2)) {
          if (f2 != 1 || wt2 != 2) continue;
          std::string key; Span val;
          if (!map_entry(sub2, key, val)) return fail(YT8M_E_BADARG, "tfrecord: malformed context map%s", "");
          if (key == "labels") parse_labels(val, lrow, num_classes);
          else if (key == "video_id") first_bytes(val, vid);
        }
      } else if (f == 2) {                               // feature_lists: FeatureLists
        Span fls = sub;
        uint32_t f2, wt2; Span sub2; uint64_t v2;
        while (next_field(fls, f2, wt2, sub2, v2)) {
          if (f2 != 1 || wt2 != 2) continue;
          std::string key; Span val;
          if (!map_entry(sub2, key, val)) return fail(YT8M_E_BADARG, "tfrecord: malformed feature_list map%s", "");
          int fi = -1;
          for (int i = 0; i < nfeat; ++i) if (key == feature_names[i]) fi = i;
          if (fi < 0) continue;
          seen[fi] = 1;
          Span fl = val;                                 // FeatureList{repeated Feature feature = 1}
          uint32_t f3, wt3; Span sub3; uint64_t v3;
          int64_t frame = 0;
          while (next_field(fl, f3, wt3, sub3, v3)) {
            if (f3 != 1 || wt3 != 2) continue;
            Span bytes;
            if (!first_bytes(sub3, bytes)) return fail(YT8M_E_BADARG, "tfrecord: frame feature '%s' is not a bytes feature", key.c_str());
            if ((bytes.e - bytes.p) != feature_sizes[fi])
              return fail(YT8M_E_SHAPE, "tfrecord: frame of '%s' has %lld bytes, expected %lld", key.c_str(),
                          (long long)(bytes.e - bytes.p), (long long)feature_sizes[fi]);
            if (frame < max_frames) memcpy(qrow + frame * D + off[fi], bytes.p, (size_t)feature_sizes[fi]);
            ++frame;
          }
          const int64_t n = frame < max_frames ? frame : max_frames;     // readers.py:181
          if (nf >= 0 && nf != n) return fail(YT8M_E_SHAPE, "tfrecord: features disagree on the number of frames%s", "");  // readers.py:239
          nf = n;
        }
      }
    }
    if (!ex.ok) return fail(YT8M_E_BADARG, "tfrecord: malformed SequenceExample%s", "");
    for (int i = 0; i < nfeat; ++i)
      if (!seen[i]) return fail(YT8M_E_BADARG, "tfrecord: feature list '%s' is missing", feature_names[i]);
    num_frames[rec] = (int32_t)(nf < 0 ? 0 : nf);
    if (video_ids) copy_id(vid, video_ids + rec * id_stride, id_stride);
    ++*n_read;
  }
  return YT8M_OK;
}

// Video-level batch: x [max_records, D] float32, labels [max_records, num_classes] uint8.
extern "C" int yt8m_tfrecord_read_video_batch(void* reader, const char* const* feature_names, const int32_t* feature_sizes,
                                              int nfeat, int64_t num_classes, int64_t max_records, float* x, uint8_t* labels,
                                              char* video_ids, int64_t id_stride, int64_t* n_read) {
  using namespace yt8m;
  YT8M_REQUIRE(reader && feature_names && feature_sizes && x && labels && n_read, YT8M_E_BADARG, "null argument");
  YT8M_REQUIRE(nfeat >= 1 && num_classes >= 1 && max_records >= 0, YT8M_E_SHAPE, "bad sizes");
  Reader* r = static_cast<Reader*>(reader);
  int64_t D = 0;
  std::vector<int64_t> off(nfeat);
  for (int i = 0; i < nfeat; ++i) { off[i] = D; D += feature_sizes[i]; }
  *n_read = 0;
  for (int64_t rec = 0; rec < max_records; ++rec) {
    const int rc = read_record(r);
    if (rc == 0) break;
    if (rc < 0) return rc;
    float* xrow = x + rec * D;
    uint8_t* lrow = labels + rec * num_classes;
    memset(lrow, 0, (size_t)num_classes);
    std::vector<char> seen(nfeat, 0);
    Span vid{nullptr, nullptr, true};
    Span ex{r->buf.data(), r->buf.data() + r->buf.size(), true};
    uint32_t f, wt; Span sub; uint64_t v;
    while (next_field(ex, f, wt, sub, v)) {
      if (f != 1 || wt != 2) continue;                   // Example{Features features = 1}
      Span feats = sub;
      uint32_t f2, wt2; Span sub2; uint64_t v2;
      while (next_field(feats, f2, wt2, sub2, v2)) {
        if (f2 != 1 || wt2 != 2) continue;
        std::string key; Span val;
        if (!map_entry(sub2, key, val)) return fail(YT8M_E_BADARG, "tfrecord: malformed feature map%s", "");
        if (key == "labels") { parse_labels(val, lrow, num_classes); continue; }
        if (key == "video_id") { first_bytes(val, vid); continue; }
        for (int i = 0; i < nfeat; ++i) {
          if (key != feature_names[i]) continue;
          const int64_t n = parse_floats(val, xrow + off[i], feature_sizes[i]);
          if (n != feature_sizes[i])                     // FixedLenFeature shape mismatch is an error in tf.parse_example
            return fail(YT8M_E_SHAPE, "tfrecord: feature '%s' has %lld floats, expected %lld", key.c_str(), (long long)n,
                        (long long)feature_sizes[i]);
          seen[i] = 1;
        }
      }
    }
    if (!ex.ok) return fail(YT8M_E_BADARG, "tfrecord: malformed Example%s", "");
    for (int i = 0; i < nfeat; ++i)
      if (!seen[i]) return fail(YT8M_E_BADARG, "tfrecord: feature '%s' is missing", feature_names[i]);
    if (video_ids) copy_id(vid, video_ids + rec * id_stride, id_stride);
    ++*n_read;
  }
  return YT8M_OK;
}

// ---- prediction dump for the ensemble stage (W/inference-pre-ensemble.py:291-308 write_to_record / get_output_feature) ----
// One tf.train.Example per video: {"video_id": bytes, "labels": int64 list = nonzero(label row), <feature_name>: float list =
// the prediction row}, framed as TFRecord.  Repeated scalars are written packed (what TF's proto3 Example encoders emit);
// map entries in the order video_id, labels, feature (readers are order-agnostic).
namespace {

void put_varint(std::string& o, uint64_t v) {
  while (v >= 0x80) { o.push_back((char)((v & 0x7f) | 0x80)); v >>= 7; }
  o.push_back((char)v);
}
void put_ld(std::string& o, int field, const std::string& payload) {
  put_varint(o, ((uint64_t)field << 3) | 2);
  put_varint(o, payload.size());
  o += payload;
}
std::string map_entry(const char* key, const std::string& feature) {
  std::string e;
  put_ld(e, 1, std::string(key));
  put_ld(e, 2, feature);
  return e;
}

}  // namespace

extern "C" int yt8m_tfrecord_write_predictions(const char* path, int64_t n, const char* video_ids, int64_t id_stride,
                                               const uint8_t* labels, const float* predictions, int64_t num_classes,
                                               const char* feature_name) {
  YT8M_REQUIRE(path && feature_name && n >= 0 && num_classes >= 0 && id_stride > 0, YT8M_E_BADARG, "bad argument");
  YT8M_REQUIRE(n == 0 || (video_ids && labels && predictions), YT8M_E_BADARG, "null operand");
  FILE* fh = fopen(path, "wb");
  if (!fh) return yt8m::fail(YT8M_E_BADARG, "yt8m_tfrecord_write_predictions: cannot open %s", path);
  int rc = YT8M_OK;
  for (int64_t r = 0; r < n && rc == YT8M_OK; ++r) {
    const char* id = video_ids + r * id_stride;
    std::string bl;                                                   // BytesList { value = id }
    put_ld(bl, 1, std::string(id, strnlen(id, (size_t)id_stride)));
    std::string f_id;
    put_ld(f_id, 1, bl);                                              // Feature.bytes_list = 1
    std::string ints;
    for (int64_t c = 0; c < num_classes; ++c)
      if (labels[r * num_classes + c]) put_varint(ints, (uint64_t)c);
    std::string il;
    put_ld(il, 1, ints);                                              // Int64List { packed value }
    std::string f_lab;
    put_ld(f_lab, 3, il);                                             // Feature.int64_list = 3
    std::string fl;
    put_ld(fl, 1, std::string(reinterpret_cast<const char*>(predictions + r * num_classes), (size_t)num_classes * 4));
    std::string f_pred;
    put_ld(f_pred, 2, fl);                                            // Feature.float_list = 2 (little-endian host)
    std::string feats;                                                // Features { map<string, Feature> feature = 1 }
    put_ld(feats, 1, map_entry("video_id", f_id));
    put_ld(feats, 1, map_entry("labels", f_lab));
    put_ld(feats, 1, map_entry(feature_name, f_pred));
    std::string ex;
    put_ld(ex, 1, feats);                                             // Example { features = 1 }
    const uint64_t len = ex.size();
    uint8_t hdr[12];
    memcpy(hdr, &len, 8);
    const uint32_t hc = mask_crc(crc32c(hdr, 8));
    memcpy(hdr + 8, &hc, 4);
    const uint32_t pc = mask_crc(crc32c(reinterpret_cast<const uint8_t*>(ex.data()), ex.size()));
    if (fwrite(hdr, 1, 12, fh) != 12 || fwrite(ex.data(), 1, ex.size(), fh) != ex.size() || fwrite(&pc, 1, 4, fh) != 4) {
      rc = yt8m::fail(YT8M_E_BADARG, "yt8m_tfrecord_write_predictions: short write to %s", path);
    }
  }
  if (fclose(fh) != 0 && rc == YT8M_OK) rc = yt8m::fail(YT8M_E_BADARG, "yt8m_tfrecord_write_predictions: close failed for %s", path);
  return rc;
}

// ---- multi-threaded shard prefetcher (the role of the reference's num_readers queue-runner threads + batch_join,
// W/train.py:199-209 / W/readers.py prepare_reader) ---------------------------------------------------------------------------
// nthreads workers pull shard indices from a shared counter, decode whole batches straight into slots of PINNED host memory
// (hipHostMalloc; plain malloc when no HIP device is present) and hand them to the consumer through a bounded ready queue.
// A batch never spans two shards (like the reference's read_up_to: short batches at shard ends).  With one thread the batch
// sequence equals the sequential reader's; with more, the order of batches across shards is scheduling-dependent (the
// reference shuffles anyway) but every record is delivered exactly once.
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

namespace {

struct Slot {
  uint8_t* q = nullptr;      // frame level: [batch, max_frames, D] uint8
  float* x = nullptr;        // video level: [batch, D] float
  int32_t* nf = nullptr;     // [batch]
  uint8_t* labels = nullptr; // [batch, num_classes]
  char* ids = nullptr;       // [batch, id_stride]
  int64_t n = 0;
};

struct Prefetcher {
  std::vector<std::string> paths;
  std::vector<std::string> names;
  std::vector<const char*> cnames;
  std::vector<int32_t> sizes;
  bool frame_level = true, check_crc = true, pinned = false;
  int64_t max_frames = 0, num_classes = 0, batch = 0, id_stride = 32, D = 0;
  std::vector<Slot> slots;
  std::deque<int> free_slots, ready;
  std::mutex mu;
  std::condition_variable cv_free, cv_ready;
  std::atomic<size_t> next_shard{0};
  int active_workers = 0;
  bool stop = false;
  int error = 0;
  std::string error_msg;
  int held = -1;              // slot currently lent to the consumer
  std::vector<std::thread> threads;
  std::vector<void*> pinned_ptrs;   // which buffers came from hipHostMalloc (freed with hipHostFree; the rest with free)
};

// Pinned while hipHostMalloc succeeds; after the first failure the remaining buffers are pageable, P->pinned (what the
// consumer is told: "every buffer is pinned") drops to false, and each buffer is released by the allocator it came from.
void* host_alloc(size_t bytes, Prefetcher* P) {
  void* p = nullptr;
  if (P->pinned && hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess && p) {
    P->pinned_ptrs.push_back(p);
    return p;
  }
  (void)hipGetLastError();
  P->pinned = false;
  return malloc(bytes);
}

void prefetch_worker(Prefetcher* P) {
  for (;;) {
    const size_t si = P->next_shard.fetch_add(1);
    if (si >= P->paths.size()) break;
    void* rd = nullptr;
    int rc = yt8m_tfrecord_open(P->paths[si].c_str(), P->check_crc ? 1 : 0, &rd);
    while (rc == YT8M_OK) {
      int s;
      {
        std::unique_lock<std::mutex> lk(P->mu);
        P->cv_free.wait(lk, [&] { return P->stop || !P->free_slots.empty(); });
        if (P->stop) break;
        s = P->free_slots.front();
        P->free_slots.pop_front();
      }
      Slot& sl = P->slots[s];
      int64_t n = 0;
      if (P->frame_level)
        rc = yt8m_tfrecord_read_frame_batch(rd, P->cnames.data(), P->sizes.data(), (int)P->sizes.size(), P->max_frames,
                                            P->num_classes, P->batch, sl.q, sl.nf, sl.labels, sl.ids, P->id_stride, &n);
      else
        rc = yt8m_tfrecord_read_video_batch(rd, P->cnames.data(), P->sizes.data(), (int)P->sizes.size(), P->num_classes, P->batch,
                                            sl.x, sl.labels, sl.ids, P->id_stride, &n);
      std::unique_lock<std::mutex> lk(P->mu);
      if (rc != YT8M_OK || n == 0) {                        // error or end of shard: the slot goes back unused
        P->free_slots.push_back(s);
        P->cv_free.notify_one();
        break;
      }
      sl.n = n;
      P->ready.push_back(s);
      P->cv_ready.notify_one();
      if (n < P->batch) break;                              // short batch = end of this shard
    }
    if (rd) yt8m_tfrecord_close(rd);
    if (rc != YT8M_OK) {
      std::unique_lock<std::mutex> lk(P->mu);
      if (!P->error) {
        P->error = rc;
        P->error_msg = yt8m_last_error();                   // thread-local message of THIS worker
      }
      P->stop = true;
      P->cv_free.notify_all();
      P->cv_ready.notify_all();
      break;
    }
    {
      std::unique_lock<std::mutex> lk(P->mu);
      if (P->stop) break;
    }
  }
  std::unique_lock<std::mutex> lk(P->mu);
  P->active_workers--;
  P->cv_ready.notify_all();
}

}  // namespace

extern "C" int yt8m_prefetch_open(const char* const* paths, int npaths, int frame_level, const char* const* feature_names,
                                  const int32_t* feature_sizes, int nfeat, int64_t max_frames, int64_t num_classes,
                                  int64_t batch, int nthreads, int queue_depth, int check_crc, void** out) {
  using namespace yt8m;
  YT8M_REQUIRE(paths && feature_names && feature_sizes && out, YT8M_E_BADARG, "null argument");
  YT8M_REQUIRE(npaths >= 1 && nfeat >= 1 && num_classes >= 1 && batch >= 1 && nthreads >= 1 && queue_depth >= 1 &&
                   (!frame_level || max_frames >= 1), YT8M_E_SHAPE, "bad sizes");
  Prefetcher* P = new Prefetcher();
  for (int i = 0; i < npaths; ++i) P->paths.push_back(paths[i]);
  for (int i = 0; i < nfeat; ++i) { P->names.push_back(feature_names[i]); P->sizes.push_back(feature_sizes[i]); P->D += feature_sizes[i]; }
  for (auto& s : P->names) P->cnames.push_back(s.c_str());
  P->frame_level = frame_level != 0;
  P->check_crc = check_crc != 0;
  P->max_frames = max_frames; P->num_classes = num_classes; P->batch = batch;
  nthreads = std::min(nthreads, npaths);
  const int nslots = queue_depth + nthreads + 1;            // ready queue + one in flight per worker + the one lent out
  P->pinned = true;
  P->slots.resize(nslots);
  for (int s = 0; s < nslots; ++s) {
    Slot& sl = P->slots[s];
    if (P->frame_level) {
      sl.q = static_cast<uint8_t*>(host_alloc((size_t)(batch * max_frames * P->D), P));
      sl.nf = static_cast<int32_t*>(host_alloc((size_t)batch * 4, P));
    } else {
      sl.x = static_cast<float*>(host_alloc((size_t)(batch * P->D) * 4, P));
    }
    sl.labels = static_cast<uint8_t*>(host_alloc((size_t)(batch * num_classes), P));
    sl.ids = static_cast<char*>(host_alloc((size_t)(batch * P->id_stride), P));
    P->free_slots.push_back(s);
  }
  P->active_workers = nthreads;
  for (int t = 0; t < nthreads; ++t) P->threads.emplace_back(prefetch_worker, P);
  *out = P;
  return YT8M_OK;
}

// Lends the next ready batch (host pointers valid until the next acquire / close).  *n = 0: every shard is exhausted.
extern "C" int yt8m_prefetch_acquire(void* handle, void** q_or_x, int32_t** num_frames, uint8_t** labels, char** video_ids,
                                     int64_t* id_stride, int64_t* n, int* pinned) {
  using namespace yt8m;
  YT8M_REQUIRE(handle && q_or_x && labels && n, YT8M_E_BADARG, "null argument");
  Prefetcher* P = static_cast<Prefetcher*>(handle);
  std::unique_lock<std::mutex> lk(P->mu);
  if (P->held >= 0) {                                       // give the previous batch's slot back to the workers
    P->free_slots.push_back(P->held);
    P->held = -1;
    P->cv_free.notify_one();
  }
  P->cv_ready.wait(lk, [&] { return !P->ready.empty() || P->active_workers == 0 || P->error; });
  if (P->error) return fail(P->error, "prefetch: %s", P->error_msg.c_str());
  if (P->ready.empty()) { *n = 0; return YT8M_OK; }
  const int s = P->ready.front();
  P->ready.pop_front();
  P->held = s;
  Slot& sl = P->slots[s];
  *q_or_x = P->frame_level ? static_cast<void*>(sl.q) : static_cast<void*>(sl.x);
  if (num_frames) *num_frames = sl.nf;
  *labels = sl.labels;
  if (video_ids) *video_ids = sl.ids;
  if (id_stride) *id_stride = P->id_stride;
  if (pinned) *pinned = P->pinned ? 1 : 0;
  *n = sl.n;
  return YT8M_OK;
}

extern "C" int yt8m_prefetch_close(void* handle) {
  Prefetcher* P = static_cast<Prefetcher*>(handle);
  if (!P) return YT8M_OK;
  {
    std::unique_lock<std::mutex> lk(P->mu);
    P->stop = true;
    P->cv_free.notify_all();
    P->cv_ready.notify_all();
  }
  for (auto& t : P->threads) t.join();
  for (auto& sl : P->slots) {
    void* ptrs[5] = {sl.q, sl.x, sl.nf, sl.labels, sl.ids};
    for (void* p : ptrs) {
      if (!p) continue;
      if (std::find(P->pinned_ptrs.begin(), P->pinned_ptrs.end(), p) != P->pinned_ptrs.end()) (void)hipHostFree(p);
      else free(p);
    }
  }
  delete P;
  return YT8M_OK;
}
