// Saved-map reader / writer (include/voxgraph_amd.h "Saved maps"): cblox submap collections as
// voxgraph writes them (voxgraph_mapper.cpp:412-417) and loads them
// (registration_test_bench.cpp:173-175, voxgraph_submap.cpp:398-415), and voxblox layer files.
// Hand-written protobuf wire format; host only.  Message schemas: vgx_mapfile_schema.h [recalled].
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <array>
#include <exception>
#include <map>
#include <string>
#include <vector>

#include "voxgraph_amd.h"
#include "vgx_mapfile_schema.h"

namespace {
using namespace vgx_schema;

thread_local std::string g_open_error;

// ---------------------------------------------------------------- wire format
struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  Reader(const uint8_t* b, const uint8_t* e) : p(b), end(e) {}
  bool done() const { return p >= end || !ok; }
  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      if (p >= end) {
        ok = false;
        return 0;
      }
      const uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7F) << shift;
      if (!(b & 0x80)) return v;
    }
    ok = false;
    return 0;
  }
  uint64_t fixed(int bytes) {
    if (end - p < bytes) {
      ok = false;
      return 0;
    }
    uint64_t v = 0;
    for (int i = 0; i < bytes; ++i) v |= (uint64_t)p[i] << (8 * i);
    p += bytes;
    return v;
  }
  Reader sub() {  // length-delimited payload
    const uint64_t n = varint();
    if (!ok || (uint64_t)(end - p) < n) {
      ok = false;
      return Reader(p, p);
    }
    Reader r(p, p + n);
    p += n;
    return r;
  }
  void skip(int wire_type) {
    switch (wire_type) {
      case 0: (void)varint(); break;
      case 1: (void)fixed(8); break;
      case 2: (void)sub(); break;
      case 5: (void)fixed(4); break;
      default: ok = false;
    }
  }
};

// a numeric field may arrive as double (wire 1), float (wire 5) or varint (wire 0)
double number(Reader& r, int wire_type) {
  if (wire_type == 1) {
    const uint64_t b = r.fixed(8);
    double d;
    std::memcpy(&d, &b, 8);
    return d;
  }
  if (wire_type == 5) {
    const uint32_t b = (uint32_t)r.fixed(4);
    float f;
    std::memcpy(&f, &b, 4);
    return f;
  }
  if (wire_type == 0) return (double)(int64_t)r.varint();
  r.skip(wire_type);
  return 0.0;
}

// number() -> integer without undefined behaviour on what a corrupt file can hold (NaN, infinities, 1e300): out of range = `bad`
int64_t to_int(double v, int64_t lo, int64_t hi, int64_t bad) {
  if (!(v >= (double)lo && v <= (double)hi)) return bad;
  return (int64_t)v;
}
constexpr int kMaxVoxelsPerSide = 256;   // (voxblox: 8 / 16 / 32; a header beyond this is not a map)

struct Writer {
  std::vector<uint8_t> b;
  void varint(uint64_t v) {
    while (v >= 0x80) {
      b.push_back((uint8_t)(v | 0x80));
      v >>= 7;
    }
    b.push_back((uint8_t)v);
  }
  void tag(int field, int wire_type) { varint((uint64_t)field << 3 | (uint64_t)wire_type); }
  void f_double(int field, double d) {
    tag(field, 1);
    uint64_t u;
    std::memcpy(&u, &d, 8);
    for (int i = 0; i < 8; ++i) b.push_back((uint8_t)(u >> (8 * i)));
  }
  void f_varint(int field, uint64_t v) {
    tag(field, 0);
    varint(v);
  }
  void f_bytes(int field, const std::vector<uint8_t>& payload) {
    tag(field, 2);
    varint(payload.size());
    b.insert(b.end(), payload.begin(), payload.end());
  }
  void f_string(int field, const char* s) {
    tag(field, 2);
    const size_t n = std::strlen(s);
    varint(n);
    b.insert(b.end(), s, s + n);
  }
};

// ---------------------------------------------------------------- file index
struct BlockRef {
  size_t offset, size;  // BlockProto payload inside the file buffer
};
struct SubmapEntry {
  vgx_map_file_submap_info info{};
  std::vector<BlockRef> tsdf, esdf;
};
}  // namespace

struct vgx_map_file_s {
  std::vector<uint8_t> data;
  std::vector<SubmapEntry> submaps;
  std::string error;
};

namespace {
int fail(vgx_map_file f, const std::string& msg) {
  if (f) f->error = msg;
  g_open_error = msg;
  return VGX_ERR_INVALID;
}

struct BlockHeader {
  int vps = 0;
  double voxel_size = 0, origin[3] = {0, 0, 0};
  bool has_data = false;
};

// Parses one BlockProto; when `words` is given the voxel_data payload is appended to it.
bool parse_block(const uint8_t* p, size_t n, BlockHeader* h, std::vector<uint32_t>* words) {
  Reader r(p, p + n);
  while (!r.done()) {
    const uint64_t key = r.varint();
    const int field = (int)(key >> 3), wt = (int)(key & 7);
    if (!r.ok) break;
    if (field == kBlockVoxelsPerSide) h->vps = (int)to_int(number(r, wt), 1, kMaxVoxelsPerSide, 0);
    else if (field == kBlockVoxelSize) h->voxel_size = number(r, wt);
    else if (field == kBlockOriginX) h->origin[0] = number(r, wt);
    else if (field == kBlockOriginY) h->origin[1] = number(r, wt);
    else if (field == kBlockOriginZ) h->origin[2] = number(r, wt);
    else if (field == kBlockHasData) h->has_data = number(r, wt) != 0;
    else if (field == kBlockVoxelData) {
      if (wt == 2) {  // packed
        Reader s = r.sub();
        if (words)
          while (!s.done()) words->push_back((uint32_t)s.varint());
        if (!s.ok) r.ok = false;
      } else if (wt == 0) {  // one element per tag
        const uint32_t v = (uint32_t)r.varint();
        if (words) words->push_back(v);
      } else {
        r.skip(wt);
      }
    } else {
      r.skip(wt);
    }
  }
  return r.ok;
}

// Layer::computeBlockIndexFromOrigin [recalled]: round(origin / block_size)
void block_index_of(const BlockHeader& h, int32_t out[3]) {
  const double bs = h.voxel_size * h.vps;
  for (int a = 0; a < 3; ++a) {
    const double q = h.origin[a] / bs;   // (a corrupt origin: NaN, infinity, 1e300 -> block 0, never an out-of-range conversion)
    out[a] = (q > -2147483000.0 && q < 2147483000.0) ? (int32_t)std::llround(q) : 0;
  }
}

bool read_message(Reader& r, BlockRef* ref, const uint8_t* base) {
  Reader s = r.sub();
  if (!r.ok) return false;
  ref->offset = (size_t)(s.p - base);
  ref->size = (size_t)(s.end - s.p);
  return true;
}

void parse_xyz(Reader s, double* out, int n) {
  while (!s.done()) {
    const uint64_t key = s.varint();
    const int field = (int)(key >> 3), wt = (int)(key & 7);
    if (field >= 1 && field <= n) out[field - 1] = number(s, wt);
    else s.skip(wt);
  }
}

bool index_collection(vgx_map_file f) {
  const uint8_t* base = f->data.data();
  Reader r(base, base + f->data.size());
  Reader head = r.sub();
  if (!r.ok) return false;
  double voxel_size = 0;
  int vps = 0;
  uint64_t n_submaps = 0;
  while (!head.done()) {
    const uint64_t key = head.varint();
    const int field = (int)(key >> 3), wt = (int)(key & 7);
    if (field == kCollectionVoxelSize) voxel_size = number(head, wt);
    else if (field == kCollectionVoxelsPerSide) vps = (int)to_int(number(head, wt), 1, kMaxVoxelsPerSide, 0);
    else if (field == kCollectionNumSubmaps) n_submaps = (uint64_t)to_int(number(head, wt), 0, (int64_t)1 << 40, 0);
    else head.skip(wt);
  }
  if (!head.ok || vps <= 0 || !(voxel_size > 0)) return false;
  for (uint64_t i = 0; i < n_submaps; ++i) {
    Reader sh = r.sub();
    if (!r.ok) return false;
    SubmapEntry e;
    e.info.voxel_size = voxel_size;
    e.info.voxels_per_side = vps;
    e.info.T_M_S[0] = 1.0;
    uint64_t n_tsdf = 0, n_esdf = 0;
    while (!sh.done()) {
      const uint64_t key = sh.varint();
      const int field = (int)(key >> 3), wt = (int)(key & 7);
      if (field == kSubmapId) e.info.id = to_int(number(sh, wt), INT64_MIN / 2, INT64_MAX / 2, 0);
      else if (field == kSubmapNumBlocks) n_tsdf = (uint64_t)to_int(number(sh, wt), 0, (int64_t)1 << 40, 0);
      else if (field == kSubmapNumEsdfBlocks) n_esdf = (uint64_t)to_int(number(sh, wt), 0, (int64_t)1 << 40, 0);
      else if (field == kSubmapTransform && wt == 2) {
        Reader t = sh.sub();
        while (!t.done()) {
          const uint64_t k2 = t.varint();
          const int f2 = (int)(k2 >> 3), w2 = (int)(k2 & 7);
          if (f2 == kTransformPosition && w2 == 2) parse_xyz(t.sub(), &e.info.T_M_S[4], 3);
          else if (f2 == kTransformRotation && w2 == 2) parse_xyz(t.sub(), &e.info.T_M_S[0], 4);
          else t.skip(w2);
        }
      } else {
        sh.skip(wt);
      }
    }
    if (!sh.ok) return false;
    for (uint64_t b = 0; b < n_tsdf; ++b) {
      BlockRef ref;
      if (!read_message(r, &ref, base)) return false;
      e.tsdf.push_back(ref);
    }
    for (uint64_t b = 0; b < n_esdf; ++b) {
      BlockRef ref;
      if (!read_message(r, &ref, base)) return false;
      e.esdf.push_back(ref);
    }
    e.info.n_tsdf_blocks = (int32_t)e.tsdf.size();
    e.info.n_esdf_blocks = (int32_t)e.esdf.size();
    f->submaps.push_back(std::move(e));
  }
  return true;
}

bool index_layer(vgx_map_file f) {
  const uint8_t* base = f->data.data();
  Reader r(base, base + f->data.size());
  const uint64_t count = r.varint();
  if (!r.ok || count == 0) return false;
  Reader head = r.sub();
  if (!r.ok) return false;
  SubmapEntry e;
  e.info.T_M_S[0] = 1.0;
  std::string type;
  while (!head.done()) {
    const uint64_t key = head.varint();
    const int field = (int)(key >> 3), wt = (int)(key & 7);
    if (field == kLayerVoxelSize) e.info.voxel_size = number(head, wt);
    else if (field == kLayerVoxelsPerSide) e.info.voxels_per_side = (int32_t)to_int(number(head, wt), 1, kMaxVoxelsPerSide, 0);
    else if (field == kLayerType && wt == 2) {
      Reader s = head.sub();
      type.assign((const char*)s.p, (size_t)(s.end - s.p));
    } else {
      head.skip(wt);
    }
  }
  if (!head.ok || e.info.voxels_per_side <= 0 || !(e.info.voxel_size > 0)) return false;
  e.info.layer_is_esdf = type == "esdf";
  std::vector<BlockRef>& dst = e.info.layer_is_esdf ? e.esdf : e.tsdf;
  for (uint64_t b = 1; b < count; ++b) {
    BlockRef ref;
    if (!read_message(r, &ref, base)) return false;
    dst.push_back(ref);
  }
  e.info.n_tsdf_blocks = (int32_t)e.tsdf.size();
  e.info.n_esdf_blocks = (int32_t)e.esdf.size();
  f->submaps.push_back(std::move(e));
  return true;
}

std::vector<uint8_t> encode_block(int vps, double voxel_size, const int32_t bi[3],
                                  const std::vector<uint32_t>& words) {
  Writer w;
  const double bs = voxel_size * vps;
  w.f_varint(kBlockVoxelsPerSide, (uint64_t)vps);
  w.f_double(kBlockVoxelSize, voxel_size);
  w.f_double(kBlockOriginX, bi[0] * bs);
  w.f_double(kBlockOriginY, bi[1] * bs);
  w.f_double(kBlockOriginZ, bi[2] * bs);
  w.f_varint(kBlockHasData, 1);
  Writer packed;
  for (uint32_t v : words) packed.varint(v);
  w.f_bytes(kBlockVoxelData, packed.b);
  return w.b;
}

void put_message(std::vector<uint8_t>& out, const std::vector<uint8_t>& msg) {
  Writer w;
  w.varint(msg.size());
  out.insert(out.end(), w.b.begin(), w.b.end());
  out.insert(out.end(), msg.begin(), msg.end());
}

uint32_t bits(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}
float from_bits(uint32_t u) {
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
}  // namespace

extern "C" {

int vgx_map_file_open(const char* path, int32_t format, vgx_map_file* out) {
  if (!path || !out) return fail(nullptr, "vgx_map_file_open: null argument");
  *out = nullptr;
  FILE* fp = std::fopen(path, "rb");
  if (!fp) return fail(nullptr, std::string("vgx_map_file_open: cannot open ") + path);
  vgx_map_file f = new vgx_map_file_s;
  std::fseek(fp, 0, SEEK_END);
  const long n = std::ftell(fp);
  std::fseek(fp, 0, SEEK_SET);
  f->data.resize(n > 0 ? (size_t)n : 0);
  const size_t got = n > 0 ? std::fread(f->data.data(), 1, (size_t)n, fp) : 0;
  std::fclose(fp);
  bool ok = got == f->data.size();
  if (ok) {
    if (format == VGX_FILE_CBLOX_COLLECTION) ok = index_collection(f);
    else if (format == VGX_FILE_VOXBLOX_LAYER) ok = index_layer(f);
    else ok = false;
  }
  if (!ok) {
    delete f;
    return fail(nullptr, std::string("vgx_map_file_open: ") + path + " is truncated or not in the expected format");
  }
  *out = f;
  return VGX_OK;
}

int vgx_map_file_close(vgx_map_file f) {
  delete f;
  return VGX_OK;
}

const char* vgx_map_file_last_error(vgx_map_file f) { return f ? f->error.c_str() : g_open_error.c_str(); }

int32_t vgx_map_file_num_submaps(vgx_map_file f) { return f ? (int32_t)f->submaps.size() : -1; }

int vgx_map_file_get_submap_info(vgx_map_file f, int32_t index, vgx_map_file_submap_info* info) {
  if (!f || !info) return VGX_ERR_INVALID;
  if (index < 0 || index >= (int32_t)f->submaps.size()) return fail(f, "vgx_map_file_get_submap_info: index out of range");
  *info = f->submaps[(size_t)index].info;
  return VGX_OK;
}

int vgx_map_file_read_submap(vgx_map_file f, int32_t index, int32_t* block_index, float* tsdf_distance,
                             float* tsdf_weight, uint8_t* tsdf_rgba, float* esdf_distance,
                             uint8_t* esdf_observed) {
  if (!f) return VGX_ERR_INVALID;
  if (index < 0 || index >= (int32_t)f->submaps.size()) return fail(f, "vgx_map_file_read_submap: index out of range");
  const SubmapEntry& e = f->submaps[(size_t)index];
  const int vps = e.info.voxels_per_side;
  const size_t vox = (size_t)vps * vps * vps;
  const uint8_t* base = f->data.data();
  std::map<std::array<int32_t, 3>, size_t> slot_of;
  std::vector<uint32_t> words;
  for (size_t b = 0; b < e.tsdf.size(); ++b) {
    BlockHeader h;
    words.clear();
    if (!parse_block(base + e.tsdf[b].offset, e.tsdf[b].size, &h, &words))
      return fail(f, "vgx_map_file_read_submap: malformed TSDF block");
    if (h.vps != vps) return fail(f, "vgx_map_file_read_submap: block voxels_per_side differs from the header");
    if (h.has_data && words.size() != vox * kTsdfWordsPerVoxel)
      return fail(f, "vgx_map_file_read_submap: TSDF block payload has the wrong length");
    int32_t bi[3];
    block_index_of(h, bi);
    slot_of[{bi[0], bi[1], bi[2]}] = b;
    if (block_index) std::memcpy(&block_index[3 * b], bi, sizeof(bi));
    for (size_t i = 0; i < vox; ++i) {
      const bool have = h.has_data;
      const uint32_t w0 = have ? words[3 * i] : 0, w1 = have ? words[3 * i + 1] : 0, w2 = have ? words[3 * i + 2] : 0;
      if (tsdf_distance) tsdf_distance[b * vox + i] = from_bits(w0);
      if (tsdf_weight) tsdf_weight[b * vox + i] = from_bits(w1);
      if (tsdf_rgba) {
        uint8_t* c = &tsdf_rgba[4 * (b * vox + i)];
        c[0] = (uint8_t)(w2 >> 24);  // r
        c[1] = (uint8_t)(w2 >> 16);  // g
        c[2] = (uint8_t)(w2 >> 8);   // b
        c[3] = (uint8_t)w2;          // a
      }
    }
  }
  if (esdf_distance) std::memset(esdf_distance, 0, e.tsdf.size() * vox * sizeof(float));
  if (esdf_observed) std::memset(esdf_observed, 0, e.tsdf.size() * vox);
  if (esdf_distance || esdf_observed) {
    for (size_t b = 0; b < e.esdf.size(); ++b) {
      BlockHeader h;
      words.clear();
      if (!parse_block(base + e.esdf[b].offset, e.esdf[b].size, &h, &words))
        return fail(f, "vgx_map_file_read_submap: malformed ESDF block");
      if (!h.has_data) continue;
      if (h.vps != vps || words.size() != vox * kEsdfWordsPerVoxel)
        return fail(f, "vgx_map_file_read_submap: ESDF block payload has the wrong length");
      int32_t bi[3];
      block_index_of(h, bi);
      auto it = slot_of.find({bi[0], bi[1], bi[2]});
      if (it == slot_of.end()) continue;  // ESDF block without a TSDF counterpart: not sampled by REG
      const size_t s = it->second;
      for (size_t i = 0; i < vox; ++i) {
        if (esdf_distance) esdf_distance[s * vox + i] = from_bits(words[2 * i]);
        if (esdf_observed) esdf_observed[s * vox + i] = (uint8_t)((words[2 * i + 1] & 0xFFu) != 0);
      }
    }
  }
  return VGX_OK;
}

int vgx_map_file_load_submap(vgx_ctx ctx, vgx_map_file f, int32_t index, vgx_submap* out) {
  if (!f || !out) return VGX_ERR_INVALID;
  if (index < 0 || index >= (int32_t)f->submaps.size()) return fail(f, "vgx_map_file_load_submap: index out of range");
  const SubmapEntry& e = f->submaps[(size_t)index];
  if (e.tsdf.empty() && !e.esdf.empty())
    return fail(f, "vgx_map_file_load_submap: an ESDF-only layer file cannot become a submap");
  const size_t vox = (size_t)e.info.voxels_per_side * e.info.voxels_per_side * e.info.voxels_per_side;
  const size_t nb = e.tsdf.size();
  if (nb > (size_t)INT32_MAX) return fail(f, "vgx_map_file_load_submap: more blocks than a submap can hold");
  std::vector<int32_t> bi;
  std::vector<float> td, tw, ed;
  std::vector<uint8_t> eo;
  const bool has_esdf = !e.esdf.empty();
  try {   // (sizes are the file's claims: no exception may cross the C boundary)
    bi.resize(3 * nb);
    td.resize(nb * vox);
    tw.resize(nb * vox);
    if (has_esdf) {
      ed.resize(nb * vox);
      eo.resize(nb * vox);
    }
  } catch (const std::exception&) {
    f->error = "vgx_map_file_load_submap: out of host memory for the submap the file describes";
    return VGX_ERR_NOMEM;
  }
  int rc = vgx_map_file_read_submap(f, index, bi.data(), td.data(), tw.data(), nullptr,
                                    has_esdf ? ed.data() : nullptr, has_esdf ? eo.data() : nullptr);
  if (rc != VGX_OK) return rc;
  rc = vgx_submap_create(ctx, (int32_t)e.info.id, (float)e.info.voxel_size, e.info.voxels_per_side, (int32_t)nb,
                         bi.data(), td.data(), tw.data(), has_esdf ? ed.data() : nullptr,
                         has_esdf ? eo.data() : nullptr, out);
  if (rc != VGX_OK) f->error = std::string("vgx_submap_create: ") + vgx_last_error(ctx);
  return rc;
}

int vgx_map_file_write(const char* path, int32_t format, double voxel_size, int32_t vps, int32_t n_submaps,
                       const vgx_map_file_submap_data* submaps) {
  if (!path || !submaps || n_submaps < 0 || vps <= 0 || !(voxel_size > 0))
    return fail(nullptr, "vgx_map_file_write: invalid argument");
  if (format == VGX_FILE_VOXBLOX_LAYER && n_submaps != 1)
    return fail(nullptr, "vgx_map_file_write: a layer file holds exactly one layer");
  const size_t vox = (size_t)vps * vps * vps;
  std::vector<uint8_t> out;
  auto tsdf_words = [&](const vgx_map_file_submap_data& s, int b) {
    std::vector<uint32_t> w(vox * kTsdfWordsPerVoxel);
    for (size_t i = 0; i < vox; ++i) {
      const size_t at = (size_t)b * vox + i;
      w[3 * i] = bits(s.tsdf_distance[at]);
      w[3 * i + 1] = bits(s.tsdf_weight[at]);
      const uint8_t* c = s.tsdf_rgba ? &s.tsdf_rgba[4 * at] : nullptr;
      w[3 * i + 2] = c ? ((uint32_t)c[3] | (uint32_t)c[2] << 8 | (uint32_t)c[1] << 16 | (uint32_t)c[0] << 24) : 0u;
    }
    return w;
  };
  auto esdf_words = [&](const vgx_map_file_submap_data& s, int b) {
    std::vector<uint32_t> w(vox * kEsdfWordsPerVoxel);
    for (size_t i = 0; i < vox; ++i) {
      const size_t at = (size_t)b * vox + i;
      w[2 * i] = bits(s.esdf_distance[at]);
      w[2 * i + 1] = s.esdf_observed[at] ? 1u : 0u;  // parent direction not kept by this library: zeros
    }
    return w;
  };
  if (format == VGX_FILE_VOXBLOX_LAYER) {
    const vgx_map_file_submap_data& s = submaps[0];
    Writer cnt;
    cnt.varint((uint64_t)s.n_blocks + 1);
    out = cnt.b;
    Writer head;
    head.f_double(kLayerVoxelSize, voxel_size);
    head.f_varint(kLayerVoxelsPerSide, (uint64_t)vps);
    head.f_string(kLayerType, "tsdf");
    put_message(out, head.b);
    for (int b = 0; b < s.n_blocks; ++b)
      put_message(out, encode_block(vps, voxel_size, &s.block_index[3 * b], tsdf_words(s, b)));
  } else if (format == VGX_FILE_CBLOX_COLLECTION) {
    Writer head;
    head.f_double(kCollectionVoxelSize, voxel_size);
    head.f_varint(kCollectionVoxelsPerSide, (uint64_t)vps);
    head.f_varint(kCollectionNumSubmaps, (uint64_t)n_submaps);
    put_message(out, head.b);
    for (int k = 0; k < n_submaps; ++k) {
      const vgx_map_file_submap_data& s = submaps[k];
      const bool has_esdf = s.esdf_distance && s.esdf_observed;
      Writer pos, rot, tr, sh;
      pos.f_double(1, s.T_M_S[4]);
      pos.f_double(2, s.T_M_S[5]);
      pos.f_double(3, s.T_M_S[6]);
      for (int a = 0; a < 4; ++a) rot.f_double(a + 1, s.T_M_S[a]);
      tr.f_bytes(kTransformPosition, pos.b);
      tr.f_bytes(kTransformRotation, rot.b);
      sh.f_varint(kSubmapId, (uint64_t)s.id);
      sh.f_varint(kSubmapNumBlocks, (uint64_t)s.n_blocks);
      sh.f_bytes(kSubmapTransform, tr.b);
      sh.f_varint(kSubmapNumEsdfBlocks, has_esdf ? (uint64_t)s.n_blocks : 0u);
      put_message(out, sh.b);
      for (int b = 0; b < s.n_blocks; ++b)
        put_message(out, encode_block(vps, voxel_size, &s.block_index[3 * b], tsdf_words(s, b)));
      if (has_esdf)
        for (int b = 0; b < s.n_blocks; ++b)
          put_message(out, encode_block(vps, voxel_size, &s.block_index[3 * b], esdf_words(s, b)));
    }
  } else {
    return fail(nullptr, "vgx_map_file_write: unknown format");
  }
  FILE* fp = std::fopen(path, "wb");
  if (!fp) return fail(nullptr, std::string("vgx_map_file_write: cannot create ") + path);
  const size_t put = std::fwrite(out.data(), 1, out.size(), fp);
  std::fclose(fp);
  return put == out.size() ? VGX_OK : fail(nullptr, "vgx_map_file_write: short write");
}

}  // extern "C"
