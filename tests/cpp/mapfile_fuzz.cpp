// Byte-level fuzzing of the saved-map reader (voxgraph_amd/csrc/vgx_mapfile.cpp: host-only code that parses files it did not
// write) under AddressSanitizer + UBSan on the CPU build (sanitizers are not available on the GPU pool): a valid cblox
// collection and a valid voxblox layer file, written by the library's own writer, are corrupted -- bytes flipped, runs
// overwritten, truncated, grown -- and opened, indexed and decoded again.  Every outcome but a memory error is acceptable:
// a clean refusal, or data (a flipped payload bit is still a valid file).  Built and run by tests/test_mapfile_cpu.py.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "voxgraph_amd.h"

// the one device entry point the reader's translation unit refers to (vgx_map_file_load_submap): not exercised here
extern "C" int vgx_submap_create(vgx_ctx, int32_t, float, int32_t, int32_t, const int32_t*, const float*, const float*, const float*,
                                 const uint8_t*, vgx_submap*) { return VGX_ERR_UNSUPPORTED; }
extern "C" const char* vgx_last_error(vgx_ctx) { return "stub"; }

static std::vector<uint8_t> slurp(const std::string& path) {
  std::vector<uint8_t> d;
  if (FILE* f = std::fopen(path.c_str(), "rb")) {
    std::fseek(f, 0, SEEK_END);
    d.resize((size_t)std::ftell(f));
    std::fseek(f, 0, SEEK_SET);
    if (std::fread(d.data(), 1, d.size(), f) != d.size()) d.clear();
    std::fclose(f);
  }
  return d;
}
static void spit(const std::string& path, const std::vector<uint8_t>& d) {
  FILE* f = std::fopen(path.c_str(), "wb");
  std::fwrite(d.data(), 1, d.size(), f);
  std::fclose(f);
}

// open + index + decode everything the file claims to hold, with buffers sized from what it claims (bounded: a header that asks
// for more than 64 Mi voxels is only indexed -- the reader must still not misbehave on it)
static int exercise(const std::string& path, int format, long* decoded) {
  vgx_map_file f = nullptr;
  if (vgx_map_file_open(path.c_str(), format, &f) != VGX_OK) return 0;
  const int32_t n = vgx_map_file_num_submaps(f);
  for (int32_t k = 0; k < n; ++k) {
    vgx_map_file_submap_info info;
    if (vgx_map_file_get_submap_info(f, k, &info) != VGX_OK) continue;
    const double vox = (double)info.voxels_per_side * info.voxels_per_side * info.voxels_per_side;
    if (info.voxels_per_side <= 0 || info.n_tsdf_blocks < 0 || vox * (double)info.n_tsdf_blocks > 64.0 * 1024 * 1024) {
      (void)vgx_map_file_read_submap(f, k, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);   // nothing to write into
      continue;
    }
    const size_t cells = (size_t)info.n_tsdf_blocks * (size_t)vox;
    std::vector<int32_t> bi(3 * (size_t)info.n_tsdf_blocks + 3);
    std::vector<float> td(cells + 1), tw(cells + 1), ed(cells + 1);
    std::vector<uint8_t> rgba(4 * cells + 4), eo(cells + 1);
    if (vgx_map_file_read_submap(f, k, bi.data(), td.data(), tw.data(), rgba.data(), ed.data(), eo.data()) == VGX_OK) ++*decoded;
  }
  (void)vgx_map_file_get_submap_info(f, n, nullptr);
  (void)vgx_map_file_get_submap_info(f, -1, nullptr);
  vgx_map_file_close(f);
  return 1;
}

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  const int rounds = argc > 2 ? std::atoi(argv[2]) : 2000;
  std::mt19937 rng(argc > 3 ? (unsigned)std::atoi(argv[3]) : 1u);
  // two valid files: a collection of two small submaps (TSDF + ESDF + colours) and a TSDF layer file
  const int vps = 8, nb = 3, vox = vps * vps * vps;
  std::vector<int32_t> bi = {0, 0, 0, 1, 0, 0, -1, 2, 5};
  std::vector<float> td((size_t)nb * vox), tw((size_t)nb * vox), ed((size_t)nb * vox);
  std::vector<uint8_t> rgba(4 * (size_t)nb * vox), eo((size_t)nb * vox);
  for (size_t i = 0; i < td.size(); ++i) {
    td[i] = 0.01f * (float)(i % 97) - 0.3f; tw[i] = (float)(i % 5); ed[i] = 0.02f * (float)(i % 31); eo[i] = (uint8_t)(i % 3 != 0);
    rgba[4 * i] = (uint8_t)i; rgba[4 * i + 1] = (uint8_t)(i >> 3); rgba[4 * i + 2] = 7; rgba[4 * i + 3] = 255;
  }
  vgx_map_file_submap_data sd[2];
  for (int k = 0; k < 2; ++k) {
    sd[k].id = 10 + k;
    const double T[7] = {1, 0, 0, 0, 0.5 * k, -1.0, 0.25};
    std::memcpy(sd[k].T_M_S, T, sizeof(T));
    sd[k].n_blocks = nb; sd[k].block_index = bi.data(); sd[k].tsdf_distance = td.data(); sd[k].tsdf_weight = tw.data();
    sd[k].tsdf_rgba = rgba.data(); sd[k].esdf_distance = k ? nullptr : ed.data(); sd[k].esdf_observed = k ? nullptr : eo.data();
  }
  const std::string good[2] = {dir + "/fuzz_good.cblox", dir + "/fuzz_good.tsdf"}, bad = dir + "/fuzz_bad.bin";
  if (vgx_map_file_write(good[0].c_str(), VGX_FILE_CBLOX_COLLECTION, 0.1, vps, 2, sd) != VGX_OK ||
      vgx_map_file_write(good[1].c_str(), VGX_FILE_VOXBLOX_LAYER, 0.1, vps, 1, sd) != VGX_OK)
    return std::printf("FAIL: writer\n"), 1;
  long opened = 0, decoded = 0, sane = 0;
  for (int fmt = 0; fmt < 2; ++fmt) {
    sane += exercise(good[fmt], fmt, &decoded);
    const std::vector<uint8_t> orig = slurp(good[fmt]);
    if (orig.empty()) return std::printf("FAIL: cannot read back %s\n", good[fmt].c_str()), 1;
    // the headers are where structure lives: most mutations go into the first kilobyte and around length prefixes
    for (int r = 0; r < rounds; ++r) {
      std::vector<uint8_t> d = orig;
      const int kind = (int)(rng() % 6);
      const size_t head = std::min<size_t>(d.size(), 1024);
      if (kind == 0) {                                            // a few byte flips near the front
        for (int q = 0, n = 1 + (int)(rng() % 4); q < n; ++q) d[rng() % head] ^= (uint8_t)(1u << (rng() % 8));
      } else if (kind == 1) {                                     // anywhere
        for (int q = 0, n = 1 + (int)(rng() % 8); q < n; ++q) d[rng() % d.size()] = (uint8_t)rng();
      } else if (kind == 2) {                                     // truncated
        d.resize(rng() % d.size());
      } else if (kind == 3) {                                     // a run overwritten with 0xff / 0x80 / 0x00 (endless varints, huge lengths)
        const size_t at = rng() % d.size(), len = 1 + rng() % 16;
        const uint8_t v = (uint8_t)(rng() % 3 == 0 ? 0xff : (rng() % 2 ? 0x80 : 0x00));
        for (size_t q = at; q < std::min(d.size(), at + len); ++q) d[q] = v;
      } else if (kind == 4) {                                     // bytes inserted
        const size_t at = rng() % d.size(), len = 1 + rng() % 32;
        std::vector<uint8_t> ins(len);
        for (auto& b : ins) b = (uint8_t)rng();
        d.insert(d.begin() + (long)at, ins.begin(), ins.end());
      } else {                                                    // the other format's reader on this file, too
        d[rng() % head] = (uint8_t)rng();
      }
      spit(bad, d);
      opened += exercise(bad, fmt, &decoded);
      if (kind == 5) opened += exercise(bad, 1 - fmt, &decoded);
    }
  }
  if (sane != 2) return std::printf("FAIL: the uncorrupted files did not open\n"), 1;
  std::printf("MAPFILE_FUZZ_OK %d corrupted files per format: %ld still opened, %ld submaps decoded\n", rounds, opened, decoded);
  return 0;
}
