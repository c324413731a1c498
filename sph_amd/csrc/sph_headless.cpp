// sph_headless: the reference's main loop (main.cpp:22-44) without OpenGL.
//
//   initOpenGL/initGeometry        -> nothing
//   initParticles                  -> the drop-in shim's initParticles (or a dam-break block)
//   while (!closeWindow) { updateParticles; display }
//                                  -> a fixed number of steps; `display` becomes a periodic dump of the
//                                     particle state (the hand-off a viewer or a checkpoint reader consumes)
//
// Two modes:
//   --mode shim      every step goes through updateParticles() on host arrays, exactly like main.cpp:33
//                    (host arrays valid after every call, as display() expects, render.cpp:240-253)
//   --mode resident  state stays in HBM between steps (sph_upload once, sph_step, sph_download at dumps)
//
// --frames-every F writes a picture of the particles every F steps (PPM, drawn on the GPU from the resident state with
// the reference's camera and shading: the counterpart of display(), render.cpp:240-260); resident mode only.
//
// --slabs K cuts the tank into K slabs along x, one slab context each on this GPU, and drives them through the library's
// own exchange (sph_slab_step_begin/_end, records moved by device copies): the call sequence of a K-GPU run, one
// process per GPU, with the cut planes steered by sph_slab_plan_next every --rebalance-every steps.  The result is the
// single context's, bit for bit.  Resident state only; dumps gather the slabs' own particles by id.
//
// Dump file: header {magic "SPHD", u32 version=1, u32 n, u32 step, sph_params (56 bytes)} = 72 bytes, then
// float32 pos[3n], float32 vel[3n].  A dump is also a checkpoint: --resume FILE continues from it.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sph_amd.h"
#include "../../include/sph_compute_shim.h"

namespace {

struct DumpHeader {
  char magic[4];
  uint32_t version, n, step;
  sph_params params;
};
static_assert(sizeof(DumpHeader) == 16 + sizeof(sph_params), "dump header layout");

bool write_dump(const std::string& path, const sph_params& p, uint32_t step, const std::vector<float>& pos,
                const std::vector<float>& vel) {
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  DumpHeader hd{{'S', 'P', 'H', 'D'}, 1u, (uint32_t)(pos.size() / 3), step, p};
  bool ok = std::fwrite(&hd, sizeof hd, 1, f) == 1 && std::fwrite(pos.data(), sizeof(float), pos.size(), f) == pos.size() &&
            std::fwrite(vel.data(), sizeof(float), vel.size(), f) == vel.size();
  std::fclose(f);
  return ok;
}

bool read_dump(const std::string& path, sph_params* p, uint32_t* step, std::vector<float>* pos, std::vector<float>* vel) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  DumpHeader hd;
  bool ok = std::fread(&hd, sizeof hd, 1, f) == 1 && std::memcmp(hd.magic, "SPHD", 4) == 0 && hd.version == 1;
  if (ok) {
    pos->resize((size_t)hd.n * 3);
    vel->resize((size_t)hd.n * 3);
    ok = std::fread(pos->data(), sizeof(float), pos->size(), f) == pos->size() &&
         std::fread(vel->data(), sizeof(float), vel->size(), f) == vel->size();
    *p = hd.params;
    *step = hd.step;
  }
  std::fclose(f);
  return ok;
}

// side^3 jittered lattice at rest spacing in the tank corner; tank = twice the block (dam break).
void dam_break_block(int side, uint64_t seed, sph_params* p, std::vector<float>* pos, std::vector<float>* vel) {
  const double d = std::cbrt((double)p->mass / (double)p->rho0);
  p->tank_size = (float)(2.0 * side * d);
  const size_t n = (size_t)side * side * side;
  pos->resize(n * 3);
  vel->assign(n * 3, 0.0f);
  uint64_t s = seed * 2862933555777941757ull + 3037000493ull;
  for (size_t i = 0; i < n; i++) {
    const size_t ijk[3] = {i / ((size_t)side * side), (i / side) % side, i % side};
    for (int a = 0; a < 3; a++) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      const double u = (double)(s >> 11) / 9007199254740992.0;  // [0,1)
      (*pos)[3 * i + a] = (float)((ijk[a] + 0.5) * d + (2.0 * u - 1.0) * 0.1 * d);
    }
  }
}

// Sloshing scene: the reference's gravity vector is a mutable global (compute.cpp:12) that a driver may turn between
// steps; here it swings about the z axis by +-15 degrees with a period of 400 steps.
void sloshing_gravity(uint32_t step, float g[3]) {
  const double ang = 15.0 * M_PI / 180.0 * std::sin(2.0 * M_PI * (double)step / 400.0);
  g[0] = (float)(10.0 * std::sin(ang));
  g[1] = (float)(-10.0 * std::cos(ang));
  g[2] = 0.0f;
}

bool write_ppm(const std::string& path, int w, int h, const std::vector<unsigned char>& rgb) {
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  std::fprintf(f, "P6\n%d %d\n255\n", w, h);
  bool ok = std::fwrite(rgb.data(), 1, rgb.size(), f) == rgb.size();
  std::fclose(f);
  return ok;
}

#define SLAB_TRY(expr)                                                            \
  do {                                                                            \
    if ((expr) != SPH_OK) {                                                       \
      std::fprintf(stderr, "sph_amd: %s: %s\n", #expr, sph_last_error());         \
      return false;                                                               \
    }                                                                             \
  } while (0)

// K slab contexts of one simulation on one GPU.
struct SlabChain {
  std::vector<sph_context*> ctx;
  std::vector<int32_t> cuts, caps;
  int gn = 0, n_total = 0;
  int64_t rebalances = 0;

  ~SlabChain() {
    for (sph_context* c : ctx)
      if (c) sph_destroy(c);
  }

  bool build(const sph_params& p, const std::vector<float>& pos, const std::vector<float>& vel, int nslabs, double capacity) {
    n_total = (int)(pos.size() / 3);
    SLAB_TRY(sph_global_cells(&p, &gn));
    std::vector<int> col(n_total);
    std::vector<int64_t> hist(gn, 0);
    for (int i = 0; i < n_total; i++) {
      SLAB_TRY(sph_column_of(&p, pos[3 * (size_t)i], &col[i]));
      hist[col[i]]++;
    }
    cuts.assign(nslabs + 1, 0);
    SLAB_TRY(sph_slab_plan_partition(hist.data(), gn, nslabs, SPH_SLAB_MIN_COLUMNS, cuts.data()));
    // a slab's particle capacity: its share, `capacity` times over (cut planes steered by work give a slab of
    // uncompressed fluid more particles than one of compressed fluid), plus its halo
    const int n_max = (int)std::min<double>(n_total, capacity * n_total / nslabs + 4096.0);
    const int rec_capacity = std::max(1 << 16, n_max / 2);
    ctx.assign(nslabs, nullptr);
    caps.assign(nslabs, 0);
    std::vector<float> sp, sv;
    std::vector<int32_t> ids;
    for (int r = 0; r < nslabs; r++) {
      sp.clear(); sv.clear(); ids.clear();
      for (int i = 0; i < n_total; i++)
        if (col[i] >= cuts[r] && col[i] < cuts[r + 1]) {
          sp.insert(sp.end(), &pos[3 * (size_t)i], &pos[3 * (size_t)i] + 3);
          sv.insert(sv.end(), &vel[3 * (size_t)i], &vel[3 * (size_t)i] + 3);
          ids.push_back(i);
        }
      if ((int)ids.size() > n_max) {
        std::fprintf(stderr, "slab %d starts with %zu particles, capacity %d: raise --slab-capacity\n", r, ids.size(), n_max);
        return false;
      }
      // room for the planes to wander without a full-width grid per slab
      caps[r] = std::min(gn, cuts[r + 1] - cuts[r] + std::max(64, gn / 3));
      SLAB_TRY(sph_slab_create(&p, n_max, 0, cuts[r], cuts[r + 1], caps[r], &ctx[r]));
      SLAB_TRY(sph_slab_upload(ctx[r], sp.data(), sv.data(), ids.data(), (int)ids.size()));
    }
    for (int r = 0; r < nslabs; r++)
      SLAB_TRY(sph_slab_attach_local(ctx[r], r > 0 ? ctx[r - 1] : nullptr, r + 1 < nslabs ? ctx[r + 1] : nullptr, rec_capacity));
    for (sph_context* c : ctx) SLAB_TRY(sph_slab_halo_begin(c));
    for (sph_context* c : ctx) SLAB_TRY(sph_slab_halo_end(c));
    return true;
  }

  bool set_params(const sph_params& p) {
    for (sph_context* c : ctx) SLAB_TRY(sph_set_params(c, &p));
    return true;
  }

  bool step() {  // begin on all, then end on all: every slab's records are on their way before any is waited for
    for (sph_context* c : ctx) SLAB_TRY(sph_slab_step_begin(c));
    for (sph_context* c : ctx) SLAB_TRY(sph_slab_step_end(c));
    return true;
  }

  // New cut planes for the records of the next step, balanced by work = sum of (9 + neighbours) per column.
  bool rebalance() {
    if (ctx.size() < 2) return true;
    std::vector<int64_t> load(gn, 0);
    std::vector<int32_t> counts(gn), work(gn);
    float shift = 0.0f;
    for (sph_context* c : ctx) {
      int first = 0, ncols = 0;
      float s = 0.0f;
      SLAB_TRY(sph_slab_load_work(c, counts.data(), work.data(), gn, &first, &ncols, &s));
      for (int k = 0; k < ncols; k++) load[first + k] = work[k];
      shift = std::max(shift, s);
    }
    std::vector<int32_t> next(cuts.size());
    float face_shift = 0.0f;
    SLAB_TRY(sph_slab_plan_next(load.data(), gn, (int)ctx.size(), cuts.data(), shift, 0.0f, caps.data(), next.data(), nullptr,
                                &face_shift));
    for (size_t r = 0; r < ctx.size(); r++) {
      SLAB_TRY(sph_slab_set_face_shift(ctx[r], face_shift));
      SLAB_TRY(sph_slab_set_next_columns(ctx[r], next[r], next[r + 1]));
    }
    cuts = next;
    rebalances++;
    return true;
  }

  // Every particle is owned by exactly one slab: its first sph_slab_last_own entries, scattered home by id.
  bool gather(std::vector<float>* pos, std::vector<float>* vel) {
    std::vector<float> sp, sv;
    std::vector<int32_t> ids;
    std::vector<unsigned char> seen(n_total, 0);
    int64_t placed = 0;
    for (sph_context* c : ctx) {
      const int have = sph_num_particles(c), own = sph_slab_last_own(c);
      sp.resize((size_t)have * 3); sv.resize((size_t)have * 3); ids.resize(have);
      int got = 0;
      SLAB_TRY(sph_slab_download(c, sp.data(), sv.data(), ids.data(), have, &got));
      for (int i = 0; i < own && i < got; i++) {
        const int id = ids[i];
        if (id < 0 || id >= n_total || seen[id]) {
          std::fprintf(stderr, "particle id %d owned twice or out of range\n", id);
          return false;
        }
        seen[id] = 1;
        placed++;
        std::memcpy(&(*pos)[3 * (size_t)id], &sp[3 * (size_t)i], 3 * sizeof(float));
        std::memcpy(&(*vel)[3 * (size_t)id], &sv[3 * (size_t)i], 3 * sizeof(float));
      }
    }
    if (placed != n_total) {
      std::fprintf(stderr, "%lld of %d particles found in the slabs\n", (long long)placed, n_total);
      return false;
    }
    return true;
  }

  bool strays(int64_t* lost, int64_t* missed) {
    *lost = *missed = 0;
    for (sph_context* c : ctx) {
      int64_t l = 0, m = 0;
      SLAB_TRY(sph_slab_lost(c, &l));
      SLAB_TRY(sph_slab_missed(c, &m));
      *lost += l;
      *missed += m;
    }
    return true;
  }
};

void usage() {
  std::puts(
      "sph_headless [--n N] [--scene random|dam|slosh] [--side S] [--steps K] [--seed X] [--mode shim|resident]\n"
      "             [--dump-every D] [--out PREFIX] [--resume FILE] [--exact]\n"
      "             [--frames-every F] [--frame-size WxH] [--camera-z Z]   pictures PREFIX_NNNNNN.ppm (resident mode)\n"
      "             [--slabs K] [--rebalance-every R] [--slab-capacity F]  K slab contexts on this GPU (resident mode)\n"
      "  random  the reference's scene (main.cpp:8, compute.cpp:108-117); dam  side^3 block in the corner of a tank twice\n"
      "  as wide; slosh  the same block under a gravity vector that swings +-15 degrees about z (period 400 steps)");
}

}  // namespace

int main(int argc, char** argv) {
  int n = 3000, steps = 100, side = 30, dump_every = 0;  // numParticles = 3000: main.cpp:8
  int nslabs = 0, rebalance_every = 2;
  double slab_capacity = 2.5;
  int frames_every = 0, frame_w = 800, frame_h = 800;    // windowWidth/windowHeight = 800: render.cpp:19-20
  float camera_z = -1.5f;                                // modelTranslation[2]: render.cpp:26
  uint64_t seed = 12345;
  bool exact = false;
  std::string scene = "random", mode = "resident", out = "sph_dump", resume;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto val = [&](const char* name) -> const char* {
      if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", name); std::exit(2); }
      return argv[++i];
    };
    if (a == "--n") n = std::atoi(val("--n"));
    else if (a == "--steps") steps = std::atoi(val("--steps"));
    else if (a == "--side") side = std::atoi(val("--side"));
    else if (a == "--seed") seed = std::strtoull(val("--seed"), nullptr, 10);
    else if (a == "--scene") scene = val("--scene");
    else if (a == "--mode") mode = val("--mode");
    else if (a == "--dump-every") dump_every = std::atoi(val("--dump-every"));
    else if (a == "--out") out = val("--out");
    else if (a == "--resume") resume = val("--resume");
    else if (a == "--exact") exact = true;
    else if (a == "--slabs") nslabs = std::atoi(val("--slabs"));
    else if (a == "--rebalance-every") rebalance_every = std::atoi(val("--rebalance-every"));
    else if (a == "--slab-capacity") slab_capacity = std::atof(val("--slab-capacity"));
    else if (a == "--frames-every") frames_every = std::atoi(val("--frames-every"));
    else if (a == "--camera-z") camera_z = (float)std::atof(val("--camera-z"));
    else if (a == "--frame-size") {
      if (std::sscanf(val("--frame-size"), "%dx%d", &frame_w, &frame_h) != 2) { usage(); return 2; }
    }
    else { usage(); return a == "--help" ? 0 : 2; }
  }

  sph_params p;
  sph_params_default(&p);
  std::vector<float> pos, vel;
  uint32_t step0 = 0;
  if (!resume.empty()) {
    if (!read_dump(resume, &p, &step0, &pos, &vel)) { std::fprintf(stderr, "cannot read %s\n", resume.c_str()); return 1; }
  } else if (scene == "dam" || scene == "slosh") {
    dam_break_block(side, seed, &p, &pos, &vel);
  } else {
    // the reference's own scene, through the drop-in initParticles (main.cpp:27)
    setenv("SPH_AMD_SEED", std::to_string(seed).c_str(), 1);  // copies, unlike putenv
    float *pp = nullptr, *vv = nullptr;
    initParticles(&pp, &vv, n);
    pos.assign(pp, pp + (size_t)n * 3);
    vel.assign(vv, vv + (size_t)n * 3);
    std::free(pp);  // main.cpp:37-39 releases with free(): the shim allocates with malloc
    std::free(vv);
  }
  if (exact) p.flags |= SPH_FLAG_EXACT_ORDER;
  n = (int)(pos.size() / 3);
  tankSize = p.tank_size;  // the shim path reads the reference's globals (main.cpp:12)
  a_gravity[0] = p.gravity[0]; a_gravity[1] = p.gravity[1]; a_gravity[2] = p.gravity[2];
  if (exact) setenv("SPH_AMD_EXACT", "1", 1);

  SlabChain chain;
  if (nslabs > 0) {
    if (mode != "resident" || frames_every > 0) { std::fprintf(stderr, "--slabs needs --mode resident and draws no pictures\n"); return 2; }
    if (!chain.build(p, pos, vel, nslabs, slab_capacity)) return 1;
  }
  sph_context* ctx = nullptr;
  if (mode == "resident" && nslabs == 0) {
    if (sph_create(&p, n, 0, &ctx) != SPH_OK || sph_upload(ctx, pos.data(), vel.data(), n) != SPH_OK) {
      std::fprintf(stderr, "sph_amd: %s\n", sph_last_error());
      return 1;
    }
  }
  if (frames_every > 0 && !ctx) { std::fprintf(stderr, "--frames-every needs --mode resident\n"); return 2; }
  sph_camera cam;
  sph_camera_default(&cam);
  cam.translation[2] = camera_z;
  std::vector<unsigned char> frame(frames_every > 0 ? (size_t)frame_w * frame_h * 3 : 0);
  auto t0 = std::chrono::steady_clock::now();
  for (int s = 1; s <= steps; s++) {
    if (scene == "slosh") {  // gravity of step number step0 + s - 1, counted from the start of the run (resumes continue)
      sloshing_gravity(step0 + (uint32_t)s - 1u, p.gravity);
      a_gravity[0] = p.gravity[0]; a_gravity[1] = p.gravity[1]; a_gravity[2] = p.gravity[2];  // what updateParticles reads
      if (ctx && sph_set_params(ctx, &p) != SPH_OK) { std::fprintf(stderr, "sph_amd: %s\n", sph_last_error()); return 1; }
      if (nslabs > 0 && !chain.set_params(p)) return 1;
    }
    if (nslabs > 0) {
      if (!chain.step()) return 1;
      if (rebalance_every > 0 && s % rebalance_every == 0 && !chain.rebalance()) return 1;
    } else if (ctx) {
      if (sph_step(ctx, 1) != SPH_OK) { std::fprintf(stderr, "sph_amd: %s\n", sph_last_error()); return 1; }
    } else {
      updateParticles(pos.data(), vel.data(), n);  // main.cpp:33
    }
    if (frames_every > 0 && s % frames_every == 0) {  // main.cpp:34, display() -> picture
      if (sph_render_frame(ctx, &cam, frame_w, frame_h, frame.data()) != SPH_OK) { std::fprintf(stderr, "sph_amd: %s\n", sph_last_error()); return 1; }
      char name[512];
      std::snprintf(name, sizeof name, "%s_%06u.ppm", out.c_str(), step0 + s);
      if (!write_ppm(name, frame_w, frame_h, frame)) { std::fprintf(stderr, "cannot write %s\n", name); return 1; }
    }
    if (dump_every > 0 && s % dump_every == 0) {  // main.cpp:34, display() -> dump
      if (ctx && sph_download(ctx, pos.data(), vel.data()) != SPH_OK) { std::fprintf(stderr, "sph_amd: %s\n", sph_last_error()); return 1; }
      if (nslabs > 0 && !chain.gather(&pos, &vel)) return 1;
      char name[512];
      std::snprintf(name, sizeof name, "%s_%06u.sphd", out.c_str(), step0 + s);
      if (!write_dump(name, p, step0 + s, pos, vel)) { std::fprintf(stderr, "cannot write %s\n", name); return 1; }
    }
  }
  if (ctx && sph_download(ctx, pos.data(), vel.data()) != SPH_OK) { std::fprintf(stderr, "sph_amd: %s\n", sph_last_error()); return 1; }
  if (nslabs > 0 && !chain.gather(&pos, &vel)) return 1;  // the downloads wait for the last step
  double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  double cx = 0, cy = 0, cz = 0;
  for (int i = 0; i < n; i++) { cx += pos[3 * i]; cy += pos[3 * i + 1]; cz += pos[3 * i + 2]; }
  std::printf("{\"n\": %d, \"steps\": %d, \"mode\": \"%s\", \"seconds\": %.6f, \"particle_updates_per_s\": %.4e, "
              "\"centre_of_mass\": [%.6f, %.6f, %.6f]",
              n, steps, mode.c_str(), sec, (double)n * steps / sec, cx / n, cy / n, cz / n);
  if (nslabs > 0) {
    int64_t lost = 0, missed = 0;
    if (!chain.strays(&lost, &missed)) return 1;
    std::printf(", \"slabs\": %d, \"rebalances\": %lld, \"lost\": %lld, \"missed\": %lld, \"cuts\": [", nslabs,
                (long long)chain.rebalances, (long long)lost, (long long)missed);
    for (size_t r = 0; r < chain.cuts.size(); r++) std::printf("%s%d", r ? ", " : "", chain.cuts[r]);
    std::printf("], \"own\": [");
    for (size_t r = 0; r < chain.ctx.size(); r++) std::printf("%s%d", r ? ", " : "", sph_slab_last_own(chain.ctx[r]));
    std::printf("]");
  }
  std::printf("}\n");
  char name[512];
  std::snprintf(name, sizeof name, "%s_final.sphd", out.c_str());
  if (!write_dump(name, p, step0 + steps, pos, vel)) { std::fprintf(stderr, "cannot write %s\n", name); return 1; }
  if (ctx) sph_destroy(ctx);
  return 0;
}
