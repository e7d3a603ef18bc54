// jade_render_cli.cpp — the repo's own C++ front end over the C-ABI HIP module.
//
// Mirrors main() of PathTrace.cu:1484-1741: scene (render_args.txt or a
// built-in configuration) -> BVH -> [boundary: jade_rt.h] -> BMP / PPM / PFM.
// The backend is loaded at run time (dlopen) so this binary has no HIP
// dependency of its own; the default is libjade_hip.so beside the executable.
#include <dlfcn.h>
#include <libgen.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "jade_host.hpp"

using namespace jadehost;

struct Api {
  void* h = nullptr;
  int (*abi_version)(void);
  const char* (*backend_name)(void);
  const char* (*last_error)(void);
  int (*scene_create)(const jade_scene_desc*, int, jade_scene**);
  void (*scene_destroy)(jade_scene*);
  int (*render)(jade_scene*, const jade_render_params*, float*, uint8_t*, jade_stats*);
};

static bool load_api(const std::string& path, Api& a) {
  a.h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!a.h) {
    fprintf(stderr, "cannot load backend %s: %s\n", path.c_str(), dlerror());
    return false;
  }
#define SYM(field, name)                                            \
  *(void**)(&a.field) = dlsym(a.h, name);                           \
  if (!a.field) { fprintf(stderr, "backend lacks %s\n", name); return false; }
  SYM(abi_version, "jade_abi_version")
  SYM(backend_name, "jade_backend_name")
  SYM(last_error, "jade_last_error")
  SYM(scene_create, "jade_scene_create")
  SYM(scene_destroy, "jade_scene_destroy")
  SYM(render, "jade_render")
#undef SYM
  if (a.abi_version() != JADE_ABI_VERSION) {
    // a stale pair would silently disagree on struct layouts (jade_stats, jade_render_params)
    fprintf(stderr, "backend %s speaks jade_rt ABI %d, this program was built against %d: rebuild both (make)\n", path.c_str(),
            a.abi_version(), JADE_ABI_VERSION);
    return false;
  }
  return true;
}

static void usage() {
  fprintf(stderr,
          "usage: jade_render (--config NAME | --args render_args.txt) [--width W --height H] [--spp N]\n"
          "                   [--out file.bmp|.ppm|.pfm] [--env sky|file.hdr] [--backend lib.so] [--device N] [--reference-walk] [--env-importance]\n"
          "  --env-importance: environment-visibility rays drawn by the sky's luminance instead of uniformly (NOT the reference's samples: the\n"
          "                    same image with less noise under a sky with a sun; the oracle backend refuses it)\n"
          "  --reference-walk: every hitBVH query walks what the reference walks (nodes_visited / tris_tested equal the oracle's);\n"
          "                    default: shadow / environment-visibility walks end at the hit that settles them - the same image, bit for bit\n"
          "  NAME: tiny, tinyjade, C1, C2, C3, C4, C5 (SURVEY.md section 8d); C3G = C3 with a DIR_REFRACT glass statue\n");
}

int main(int argc, char** argv) {
  std::string config, args_file, out = "RenderResultHip.bmp", backend, env;
  int width = 0, height = 0, spp = 0, device = 0;
  bool reference_walk = false;
  bool env_importance = false;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto need = [&](const char* what) -> const char* {
      if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", what); exit(2); }
      return argv[++i];
    };
    if (a == "--config") config = need("--config");
    else if (a == "--args") args_file = need("--args");
    else if (a == "--width") width = atoi(need("--width"));
    else if (a == "--height") height = atoi(need("--height"));
    else if (a == "--spp") spp = atoi(need("--spp"));
    else if (a == "--out") out = need("--out");
    else if (a == "--env") env = need("--env");
    else if (a == "--backend") backend = need("--backend");
    else if (a == "--device") device = atoi(need("--device"));
    else if (a == "--reference-walk") reference_walk = true;
    else if (a == "--env-importance") env_importance = true;
    else { usage(); return 2; }
  }
  if (config.empty() == args_file.empty()) { usage(); return 2; }
  if (backend.empty()) {
    char self[4096];
    ssize_t n = readlink("/proc/self/exe", self, sizeof self - 1);
    self[n > 0 ? n : 0] = 0;
    backend = std::string(dirname(self)) + "/libjade_hip.so";
  }

  SceneBuilder builder;
  Config cfg;
  std::string err;
  if (!config.empty()) {
    if (!make_config(config, builder, cfg, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
  } else {
    RenderArgs ra;
    if (!read_render_args(args_file, ra, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    std::string dir = args_file;
    size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
    for (const RenderArgsObject& o : ra.objects) {
      Mesh m;
      std::string f = (!o.file.empty() && o.file[0] == '/') ? o.file : dir + o.file;
      if (!load_obj(f, m, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
      builder.add_mesh(m, o.material, o.trans, o.normalize);
    }
    memcpy(cfg.eye, ra.eye, sizeof cfg.eye);
    memcpy(cfg.camera, ra.camera, sizeof cfg.camera);
    cfg.width = cfg.height = 1024;  // the reference's -DLARGE size, PathTrace.cu:25-26
    cfg.spp = 64;
    if (env.empty()) env = "sky";
  }
  if (env == "sky") builder.set_env(make_env_sky(1024, 512));
  else if (!env.empty()) {
    EnvMap e;
    if (!load_hdr(env, e, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    builder.set_env(std::move(e));
  }
  if (width > 0) cfg.width = width;
  if (height > 0) cfg.height = height;
  if (spp > 0) cfg.spp = spp;

  printf("Model load done:  %d Triangles.\n", builder.triangle_count());
  BuiltScene scene = builder.build(8);
  printf("BVH Build done: %zu nodes, depth %d, %.2f s.\n", scene.nodes.size(), scene.bvh_depth, scene.build_seconds);

  Api api;
  if (!load_api(backend, api)) return 1;
  jade_scene_desc desc = scene.desc();
  jade_scene* dev = nullptr;
  if (api.scene_create(&desc, device, &dev) != JADE_OK) { fprintf(stderr, "scene: %s\n", api.last_error()); return 1; }
  jade_render_params rp;
  memset(&rp, 0, sizeof rp);
  rp.width = cfg.width; rp.height = cfg.height; rp.spp = cfg.spp;
  memcpy(rp.eye, cfg.eye, sizeof rp.eye);
  memcpy(rp.camera, cfg.camera, sizeof rp.camera);
  rp.tile_nranks = 1;
  rp.device_id = device;
  rp.walk = reference_walk ? JADE_WALK_REFERENCE : JADE_WALK_EARLY_EXIT;
  rp.env_sampling = env_importance ? JADE_ENV_IMPORTANCE : JADE_ENV_REFERENCE;  // (non-parity: another estimator of the same image; HIP backend only)
  std::vector<float> rgb((size_t)3 * rp.width * rp.height);
  std::vector<uint8_t> bgr((size_t)3 * rp.width * rp.height);
  jade_stats st;
  memset(&st, 0, sizeof st);
  printf("Start... %dx%d, %d spp on %s\n", rp.width, rp.height, rp.spp, api.backend_name());
  if (api.render(dev, &rp, rgb.data(), bgr.data(), &st) != JADE_OK) { fprintf(stderr, "render: %s\n", api.last_error()); return 1; }
  api.scene_destroy(dev);
  double rays = (double)(st.rays_primary + st.rays_secondary);
  printf("{\"rays\": %.0f, \"kernel_ms\": %.3f, \"mray_per_s\": %.3f, \"rays_primary\": %llu, \"rays_secondary\": %llu, "
         "\"nodes_visited\": %llu, \"tris_tested\": %llu, \"shaded_hits\": %llu, \"samples\": %llu}\n",
         rays, st.kernel_ms, rays / st.kernel_ms / 1e3, (unsigned long long)st.rays_primary, (unsigned long long)st.rays_secondary,
         (unsigned long long)st.nodes_visited, (unsigned long long)st.tris_tested, (unsigned long long)st.shaded_hits,
         (unsigned long long)st.samples);
  bool ok;
  size_t dot = out.find_last_of('.');
  std::string ext = dot == std::string::npos ? "" : out.substr(dot);
  if (ext == ".pfm") ok = write_pfm(out, rgb.data(), rp.width, rp.height);
  else if (ext == ".ppm") ok = write_ppm(out, bgr.data(), rp.width, rp.height);
  else ok = write_bmp(out, bgr.data(), rp.width, rp.height);
  if (!ok) { fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
  printf("wrote %s\n", out.c_str());
  return 0;
}
