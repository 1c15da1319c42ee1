// Per-circuit quotient kernels compiled at System::new (program compilation cache keyed by the program).
// The reference interprets the node vector for every row (ConstraintGraph::sweep_range, /root/reference/src/eval.rs:67-106)
// and so does quotient_k (quotient.hip), paying an instruction fetch, a switch and two LDS round trips per node. Here
// the compiled circuit is printed as straight-line HIP - one local per node, the logUp constraints
// (src/lookup.rs:152-208) and the alpha fold (src/prover.rs:866-962) unrolled with literal column indices - and compiled
// for gfx950 with hiprtc; the register allocator then does what the slot file did by hand. hiprtc is loaded with
// dlopen: without it, or when a program is too large to be worth compiling, the interpreter kernel keeps doing the
// job (same results either way; MSAMD_NO_JIT=1 forces it). Code objects are cached per process by source hash and,
// across processes, in MSAMD_JIT_CACHE (default: .jit_cache next to the library).
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <sstream>

#include "bb_dev.h"
#include "host.h"
#include "lookup_params.h"
#include "quotient_params.h"

namespace msamd {

namespace {

bool inline_tables_fit(size_t n_zeros, size_t n_lookups, size_t quotient_degree) {
  // constraint count = user roots + 2 per lookup (2 for the pass-through when there is none): src/lookup.rs:90-99
  return n_zeros + 2 * std::max<size_t>(n_lookups, 1) <= QP_INLINE_ALPHA && quotient_degree <= 8;
}

std::string circuit_source(const std::vector<PNode>& nodes, const std::vector<uint32_t>& zeros,
                           const std::vector<std::pair<uint32_t, std::vector<uint32_t>>>& lookups, bool inl) {
  const size_t nn = nodes.size();
  const char* AR = "p.alpha_rev";  // device memory: the challenges never travel in the argument block
  (void)inl;
  std::vector<char> needed(nn, 0);
  for (auto z : zeros) needed[z] = 1;
  for (auto& l : lookups) {
    needed[l.first] = 1;
    for (auto a : l.second) needed[a] = 1;
  }
  for (size_t i = nn; i-- > 0;) {
    if (!needed[i]) continue;
    const PNode& n = nodes[i];
    if (n.kind == OP_ADD || n.kind == OP_SUB || n.kind == OP_MUL) needed[n.a] = needed[n.b] = 1;
    if (n.kind == OP_NEG) needed[n.a] = 1;
  }
  std::ostringstream o;
  o << "#include \"quotient_params.h\"\nusing namespace msamd;\n"
       "extern \"C\" __global__ __launch_bounds__(256) void quotient_jit(QParams p) {\n"
       "  const size_t t = blockIdx.x * size_t(blockDim.x) + threadIdx.x;\n"
       "  const unsigned lognq = p.log_n + p.log_q;\n"
       "  const size_t nq = size_t(1) << lognq;\n"
       "  if (t >= nq) return;\n"
       "  const u32 i = bitrev32((u32)t, lognq);\n"
       "  const u32 inext = (i + (1u << p.log_q)) & (u32)(nq - 1);\n"
       "  const size_t tn = bitrev32(inext, lognq);\n"
       "  const u32 e = i << (TW_LOG - lognq);\n"
       "  const u64 x = gl_mul_small(gl_mul(p.t1[e >> TW_HALF], p.t0[e & ((1u << TW_HALF) - 1)]), 7);\n"
       "  const u32 qi = i & ((1u << p.log_q) - 1);\n"
    << (inl ? "  const u64 zh = p.zh_in[qi & 7];\n" : "  const u64 zh = p.zh[qi];\n") <<
       "  const u64 d_first = gl_sub(x, 1), d_last = gl_sub(x, p.g_inv);\n"
       "  const u64 inv_both = gl_inv(gl_mul(d_first, d_last));\n"
       "  const u64 is_first = gl_mul(zh, gl_mul(inv_both, d_last));\n"
       "  const u64 is_last = gl_mul(zh, gl_mul(inv_both, d_first));\n"
       "  const u64 is_trans = d_last;\n"
       "  (void)is_first; (void)is_last; (void)is_trans; (void)tn;\n";
  for (size_t i = 0; i < nn; i++) {
    if (!needed[i]) continue;
    const PNode& n = nodes[i];
    o << "  const u64 v" << i << " = ";
    switch (n.kind) {
      case OP_CONST: o << n.a << "ULL"; break;
      case OP_VAR: {
        const char* src = n.source == 1 ? "s1" : n.source == 0 ? "pre" : "s2";
        o << "p." << src << "[size_t(" << n.a << ") * p." << src << "_h + " << (n.offset ? "tn" : "t") << "]";
        break;
      }
      case OP_PUBLIC: o << "p.dyn->publics[" << n.a << "]"; break;
      case OP_IS_FIRST: o << "is_first"; break;
      case OP_IS_LAST: o << "is_last"; break;
      case OP_IS_TRANS: o << "is_trans"; break;
      case OP_ADD: o << "gl_add(v" << n.a << ", v" << n.b << ")"; break;
      case OP_SUB: o << "gl_sub(v" << n.a << ", v" << n.b << ")"; break;
      case OP_MUL: o << "gl_mul(v" << n.a << ", v" << n.b << ")"; break;
      default: o << "gl_neg(v" << n.a << ")"; break;
    }
    o << ";\n";
  }
  o << "  GlAccS fa0, fa1;\n  accs_init(fa0);\n  accs_init(fa1);\n";
  size_t ci = 0;
  for (auto z : zeros) {
    o << "  { const E2 a = " << AR << "[" << ci << "]; accs_mad(fa0, v" << z << ", a.c0); accs_mad(fa1, v" << z << ", a.c1); }\n";
    ci++;
  }
  o << "  const u64 beta0 = p.dyn->publics[0], beta1 = p.dyn->publics[1];\n"
       "  const u64 inj0 = gl_mul(is_last, p.dyn->delta_scaled[0]), inj1 = gl_mul(is_last, p.dyn->delta_scaled[1]);\n"
       "  (void)beta0; (void)beta1;\n";
  auto fold2 = [&](const std::string& c0, const std::string& c1) {
    o << "  { const E2 a = " << AR << "[" << ci << "], b = " << AR << "[" << ci + 1 << "];\n"
      << "    accs_mad(fa0, " << c0 << ", a.c0); accs_mad(fa0, " << c1 << ", b.c0);\n"
      << "    accs_mad(fa1, " << c0 << ", a.c1); accs_mad(fa1, " << c1 << ", b.c1); }\n";
    ci += 2;
  };
  const size_t L = lookups.size();
  if (L == 0) {
    o << "  const u64 pt0 = gl_add(gl_sub(p.s2[tn], p.s2[t]), inj0);\n"
         "  const u64 pt1 = gl_add(gl_sub(p.s2[p.s2_h + tn], p.s2[p.s2_h + t]), inj1);\n";
    fold2("pt0", "pt1");
  } else {
    o << "  const u64 acc0_0 = p.s2[t], acc1_0 = p.s2[p.s2_h + t];\n";
    for (size_t j = 0; j < L; j++) {
      const auto& l = lookups[j];
      // running-sum columns: source = column pair j, target = pair j + 1, or the next row's first pair (+ injection)
      if (j + 1 < L)
        o << "  const u64 acc0_" << j + 1 << " = p.s2[size_t(" << 2 * j + 2 << ") * p.s2_h + t], acc1_" << j + 1 << " = p.s2[size_t("
          << 2 * j + 3 << ") * p.s2_h + t];\n";
      else
        o << "  const u64 acc0_" << j + 1 << " = gl_add(p.s2[tn], inj0), acc1_" << j + 1 << " = gl_add(p.s2[p.s2_h + tn], inj1);\n";
      const size_t na = l.second.size();
      o << "  u64 f0_" << j << ", f1_" << j << ";\n  {\n";
      if (na <= 32) {
        o << "    GlAcc g0, g1;\n    acc_init(g0);\n    acc_init(g1);\n";
        for (size_t k = 0; k < na; k++)
          o << "    acc_mad(g0, v" << l.second[k] << ", p.dyn->gpow[" << k << "].c0); acc_mad(g1, v" << l.second[k] << ", p.dyn->gpow[" << k
            << "].c1);\n";
        o << "    f0_" << j << " = acc_reduce(g0);\n    f1_" << j << " = acc_reduce(g1);\n";
      } else {
        o << "    u64 h0 = 0, h1 = 0, g0, g1;\n";
        for (size_t k = na; k-- > 0;)
          o << "    mul2(h0, h1, p.dyn->publics[2], p.dyn->publics[3], g0, g1); h0 = gl_add(g0, v" << l.second[k] << "); h1 = g1;\n";
        o << "    f0_" << j << " = h0;\n    f1_" << j << " = h1;\n";
      }
      o << "  }\n  u64 c0_" << j << ", c1_" << j << ";\n"
        << "  mul2(gl_add(f0_" << j << ", beta0), gl_add(f1_" << j << ", beta1), gl_sub(acc0_" << j + 1 << ", acc0_" << j << "), gl_sub(acc1_"
        << j + 1 << ", acc1_" << j << "), c0_" << j << ", c1_" << j << ");\n";
      std::ostringstream c0;
      c0 << "gl_sub(c0_" << j << ", v" << l.first << ")";
      std::ostringstream c1;
      c1 << "c1_" << j;
      fold2(c0.str(), c1.str());
    }
  }
  o << (inl ? "  const u64 iv = p.zh_inv_in[qi & 7];\n" : "  const u64 iv = p.zh_inv[qi];\n")
    <<
       "  p.out[t] = gl_mul(accs_reduce(fa0), iv);\n"
       "  p.out[nq + t] = gl_mul(accs_reduce(fa1), iv);\n}\n";
  return o.str();
}

// The BabyBear / Ext4 configuration's quotient kernel (bb_kernels.hip quotient_k is the interpreter it replaces: node
// values in a slot file in global memory, nodes x rows words). Same structure as above over 31-bit Montgomery words:
// selectors, the node program as locals, user constraints, logUp constraints as E4 products (src/lookup.rs:152-256),
// alpha fold, division by the vanishing polynomial.
std::string bb_circuit_source(const std::vector<PNode>& nodes, const std::vector<uint32_t>& zeros,
                              const std::vector<std::pair<uint32_t, std::vector<uint32_t>>>& lookups) {
  const size_t nn = nodes.size();
  std::vector<char> needed(nn, 0);
  for (auto z : zeros) needed[z] = 1;
  for (auto& l : lookups) {
    needed[l.first] = 1;
    for (auto a : l.second) needed[a] = 1;
  }
  for (size_t i = nn; i-- > 0;) {
    if (!needed[i]) continue;
    const PNode& n = nodes[i];
    if (n.kind == OP_ADD || n.kind == OP_SUB || n.kind == OP_MUL) needed[n.a] = needed[n.b] = 1;
    if (n.kind == OP_NEG) needed[n.a] = 1;
  }
  std::ostringstream o;
  o << "#include \"bb_quotient_params.h\"\nusing namespace msbb;\n"
       "extern \"C\" __global__ __launch_bounds__(256) void bb_quotient_jit(QuotArgs p) {\n"
       "  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;\n"
       "  if (t >= p.rows) return;\n"
       "  const unsigned log_big = p.log_n + p.log_q;\n"
       "  const size_t N = size_t(1) << log_big, q = size_t(1) << p.log_q;\n"
       "  const size_t st = p.row0 + t;\n"
       "  const size_t i = quot_bitrev(st, log_big);\n"
       "  const size_t st_next = quot_bitrev((i + q) & (N - 1), log_big);\n"
       "  const u32 x = bb_mul(p.g, bb_pow(p.w_big, i));\n"
       "  const u32 zh = bb_sub(bb_mul(p.g_pow_n, bb_pow(p.w_q, i & (q - 1))), BB_R1);\n"
       "  const u32 d1 = bb_sub(x, BB_R1), d2 = bb_sub(x, p.gn_inv);\n"
       "  const u32 d12 = bb_mul(d1, d2);\n"
       "  const u32 all_inv = bb_inv(bb_mul(d12, zh));\n"
       "  const u32 inv_zh = bb_mul(all_inv, d12);\n"
       "  const u32 inv12 = bb_mul(all_inv, zh);\n"
       "  const u32 is_first = bb_mul(zh, bb_mul(inv12, d2)), is_last = bb_mul(zh, bb_mul(inv12, d1)), is_trans = d2;\n"
       "  (void)is_first; (void)is_last; (void)is_trans; (void)st_next;\n";
  for (size_t i = 0; i < nn; i++) {
    if (!needed[i]) continue;
    const PNode& n = nodes[i];
    o << "  const u32 v" << i << " = ";
    switch (n.kind) {
      case OP_CONST: o << msbb::bb_to_monty((uint32_t)n.a) << "u"; break;
      case OP_VAR: {
        const char* src = n.source == 1 ? "s1" : n.source == 0 ? "pre" : "s2";
        o << "p." << src << "[size_t(" << n.a << ") * p." << src << "_ld + " << (n.offset ? "st_next" : "st") << "]";
        break;
      }
      case OP_PUBLIC: o << "p.publics[" << n.a << "]"; break;
      case OP_IS_FIRST: o << "is_first"; break;
      case OP_IS_LAST: o << "is_last"; break;
      case OP_IS_TRANS: o << "is_trans"; break;
      case OP_ADD: o << "bb_add(v" << n.a << ", v" << n.b << ")"; break;
      case OP_SUB: o << "bb_sub(v" << n.a << ", v" << n.b << ")"; break;
      case OP_MUL: o << "bb_mul(v" << n.a << ", v" << n.b << ")"; break;
      default: o << "bb_neg(v" << n.a << ")"; break;
    }
    o << ";\n";
  }
  o << "  E4 acc = e4_zero();\n";
  size_t cj = 0;
  for (auto z : zeros) o << "  acc = e4_add(acc, e4_mul_base(p.apow[" << cj++ << "], v" << z << "));\n";
  o << "  const E4 beta = E4{{p.publics[0], p.publics[1], p.publics[2], p.publics[3]}};\n"
       "  const E4 gamma = E4{{p.publics[4], p.publics[5], p.publics[6], p.publics[7]}};\n"
       "  (void)beta; (void)gamma;\n"
       "  const E4 inj = E4{{bb_mul(is_last, p.delta[0]), bb_mul(is_last, p.delta[1]), bb_mul(is_last, p.delta[2]), bb_mul(is_last, p.delta[3])}};\n";
  auto s2_at = [&](const char* row, size_t slot) {
    std::ostringstream e;
    e << "E4{{";
    for (int k = 0; k < 4; k++) e << (k ? ", " : "") << "p.s2[size_t(" << 4 * slot + k << ") * p.s2_ld + " << row << "]";
    e << "}}";
    return e.str();
  };
  auto fold4 = [&](const std::string& c) {
    for (int k = 0; k < 4; k++) o << "  acc = e4_add(acc, e4_mul_base(p.apow[" << cj++ << "], " << c << ".c[" << k << "]));\n";
  };
  const size_t L = lookups.size();
  if (L == 0) {
    o << "  const E4 pt = e4_add(e4_sub(" << s2_at("st_next", 0) << ", " << s2_at("st", 0) << "), inj);\n";
    fold4("pt");
  } else {
    o << "  const E4 run0 = " << s2_at("st", 0) << ";\n";
    for (size_t j = 0; j < L; j++) {
      const auto& l = lookups[j];
      if (j + 1 < L)
        o << "  const E4 run" << j + 1 << " = " << s2_at("st", j + 1) << ";\n";
      else
        o << "  const E4 run" << j + 1 << " = e4_add(" << s2_at("st_next", 0) << ", inj);\n";
      o << "  E4 c" << j << ";\n  {\n    E4 f = e4_zero();\n";
      for (size_t k = l.second.size(); k-- > 0;) o << "    f = e4_mul(f, gamma); f.c[0] = bb_add(f.c[0], v" << l.second[k] << ");\n";
      o << "    c" << j << " = e4_mul(e4_add(f, beta), e4_sub(run" << j + 1 << ", run" << j << "));\n"
        << "    c" << j << ".c[0] = bb_sub(c" << j << ".c[0], v" << l.first << ");\n  }\n";
      std::ostringstream c;
      c << "c" << j;
      fold4(c.str());
    }
  }
  o << "  acc = e4_mul_base(acc, inv_zh);\n"
       "  for (int k = 0; k < 4; k++) p.out[(size_t)k * p.out_ld + i] = acc.c[k];\n}\n";
  return o.str();
}

// ---- hiprtc through dlopen (no link-time dependency: a box without it still runs the interpreter)
struct Rtc {
  void* lib = nullptr;
  int (*create)(void**, const char*, const char*, int, const char**, const char**) = nullptr;
  int (*compile)(void*, int, const char**) = nullptr;
  int (*log_size)(void*, size_t*) = nullptr;
  int (*get_log)(void*, char*) = nullptr;
  int (*code_size)(void*, size_t*) = nullptr;
  int (*get_code)(void*, char*) = nullptr;
  int (*destroy)(void**) = nullptr;
  int version = 0;  // major * 1000 + minor: part of the cache key
  bool ok = false;
  Rtc() {
    for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) return;
    create = (decltype(create))dlsym(lib, "hiprtcCreateProgram");
    compile = (decltype(compile))dlsym(lib, "hiprtcCompileProgram");
    log_size = (decltype(log_size))dlsym(lib, "hiprtcGetProgramLogSize");
    get_log = (decltype(get_log))dlsym(lib, "hiprtcGetProgramLog");
    code_size = (decltype(code_size))dlsym(lib, "hiprtcGetCodeSize");
    get_code = (decltype(get_code))dlsym(lib, "hiprtcGetCode");
    destroy = (decltype(destroy))dlsym(lib, "hiprtcDestroyProgram");
    ok = create && compile && log_size && get_log && code_size && get_code && destroy;
    if (auto ver = (int (*)(int*, int*))dlsym(lib, "hiprtcVersion")) {
      int major = 0, minor = 0;
      if (ver(&major, &minor) == 0) version = major * 1000 + minor;
    }
  }
};

std::string library_dir() {
  Dl_info info;
  if (dladdr((void*)&library_dir, &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    size_t k = p.rfind('/');
    return k == std::string::npos ? "." : p.substr(0, k);
  }
  return ".";
}

std::mutex g_mu;
std::map<std::string, std::vector<char>> g_code;  // key digest -> code object (empty = compilation failed, do not retry)

bool read_file(const std::string& path, std::vector<char>& out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize(n > 0 ? (size_t)n : 0);
  bool ok = n > 0 && fread(out.data(), 1, (size_t)n, f) == (size_t)n;
  fclose(f);
  return ok;
}

// the architecture the code objects are built for: that of the current device (gfx950 on MI355X)
std::string device_arch() {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    (void)hipGetLastError();
    return "gfx950";
  }
  std::string a = prop.gcnArchName;  // "gfx950:sramecc+:xnack-"
  const size_t k = a.find(':');
  return k == std::string::npos ? a : a.substr(0, k);
}

// A cached code object is only ever loaded when it was built from exactly this program: the key is the BLAKE3 digest of the
// source, the headers it includes, the target architecture, the hiprtc version and the compiler options, and the file
// carries that digest in its header (a stale, foreign or truncated file is ignored and rebuilt). The cache directory
// is private to the user.
const char CACHE_MAGIC[8] = {'M', 'S', 'J', 'C', '0', '0', '0', '2'};

const std::vector<char>* code_object(const std::string& src) {
  static Rtc rtc;
  const std::string dir = library_dir();
  const std::string arch = device_arch();
  const std::string inc = "-I" + dir + "/csrc";
  const std::string arch_opt = "--offload-arch=" + arch;
  const char* opts[] = {arch_opt.c_str(), "-O3", "-std=c++17", inc.c_str()};
  std::string key = arch + '\0' + std::to_string(rtc.version) + '\0' + "-O3 -std=c++17" + '\0';
  // the headers are part of the program: a change to the field arithmetic must not reuse old code objects
  for (const char* hdr : {"/csrc/gl_dev.h", "/csrc/quotient_params.h", "/csrc/lookup_params.h", "/csrc/bb_dev.h", "/csrc/bb_quotient_params.h"}) {
    std::vector<char> t;
    if (read_file(dir + hdr, t)) key.append(t.begin(), t.end());
    key += '\0';
  }
  key += src;
  uint8_t dg[32];
  blake3_host(reinterpret_cast<const uint8_t*>(key.data()), key.size(), dg);
  char hex[65];
  for (int i = 0; i < 32; i++) snprintf(hex + 2 * i, 3, "%02x", dg[i]);
  const std::string id(hex, 64);
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_code.find(id);
  if (it != g_code.end()) return it->second.empty() ? nullptr : &it->second;
  std::vector<char>& slot = g_code[id];
  const char* env = getenv("MSAMD_JIT_CACHE");
  const std::string cache_dir = env ? env : dir + "/.jit_cache";
  const std::string path = cache_dir + "/q_" + id.substr(0, 32) + ".co";
  {
    std::vector<char> file;
    if (read_file(path, file) && file.size() > 40 && memcmp(file.data(), CACHE_MAGIC, 8) == 0 && memcmp(file.data() + 8, dg, 32) == 0) {
      slot.assign(file.begin() + 40, file.end());
      return &slot;
    }
  }
  slot.clear();
  if (!rtc.ok) return nullptr;
  void* prog = nullptr;
  if (rtc.create(&prog, src.c_str(), "quotient_jit.hip", 0, nullptr, nullptr) != 0) return nullptr;
  const int rc = rtc.compile(prog, 4, opts);
  if (rc != 0) {
    size_t ls = 0;
    rtc.log_size(prog, &ls);
    std::string log(ls, 0);
    if (ls) rtc.get_log(prog, &log[0]);
    fprintf(stderr, "[msamd] hiprtc could not compile a circuit kernel (the generic kernel is used instead):\n%s\n", log.c_str());
    rtc.destroy(&prog);
    return nullptr;
  }
  size_t cs = 0;
  rtc.code_size(prog, &cs);
  slot.resize(cs);
  rtc.get_code(prog, slot.data());
  rtc.destroy(&prog);
  // best-effort disk cache: write to a private name, then rename (several ranks may compile the same program)
  mkdir(cache_dir.c_str(), 0700);
  const std::string tmp = path + "." + std::to_string((long)getpid());
  if (FILE* f = fopen(tmp.c_str(), "wb")) {
    const bool ok = fwrite(CACHE_MAGIC, 1, 8, f) == 8 && fwrite(dg, 1, 32, f) == 32 && fwrite(slot.data(), 1, slot.size(), f) == slot.size();
    fclose(f);
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) remove(tmp.c_str());
  }
  return &slot;
}

}  // namespace

JitKernel::~JitKernel() {
  if (module) (void)hipModuleUnload((hipModule_t)module);
}

namespace {
// module + function from a code object; leaves `out` empty on any failure
void load_kernel(const std::vector<char>* co, const char* name, JitKernel& out) {
  if (!co) return;
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
  if (hipModuleLoadData(&mod, co->data()) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  if (hipModuleGetFunction(&fn, mod, name) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipModuleUnload(mod);
    return;
  }
  out.module = mod;
  out.function = fn;
}

// Stage-2 terms pass (LookupValues::stage_2_traces, /root/reference/src/lookup.rs:472-543, up to the running sum) for one
// list of argument counts: the row's multiplicities and arguments are loaded up front (16-byte loads where the row
// stride allows), every fingerprint is an unrolled lazy dot product with the gamma powers, the messages are inverted
// 16 at a time with one base-field inversion.
// groups of 16 lookups (one batch inversion each); with three or more of them a WAVE takes a group (stage2_grouped_source)
static unsigned stage2_groups(size_t n_lookups) {
  const size_t g = (n_lookups + 15) / 16;
  return g >= 3 && g <= 16 && !getenv("MSAMD_NO_STAGE2_GROUPS") ? (unsigned)g : 0u;
}

// Many lookups per row (the reference's BLAKE3 compression circuit: 73, with 729 arguments): one thread per row walks five
// batches one after the other - 0.36 ms on that circuit's 512 rows. The batches are independent, so a workgroup is 64 rows x G
// waves and wave g does batch g of its 64 rows (uniform control flow per wave); the row sums meet in LDS (field addition is
// exact: the order of the partial sums does not matter).
std::string stage2_grouped_source(const std::vector<uint32_t>& counts, unsigned G) {
  const size_t L = counts.size();
  size_t aw = 0;
  std::vector<size_t> offs;
  for (auto c : counts) {
    offs.push_back(aw);
    aw += c;
  }
  std::ostringstream o;
  o << "#include \"lookup_params.h\"\nusing namespace msamd;\n"
       "extern \"C\" __global__ __launch_bounds__(" << 64 * G << ") void stage2_terms_jit(Stage2Params p) {\n"
       "  __shared__ E2 part[" << G << "][64];\n"
       "  const unsigned lane = threadIdx.x & 63, grp = threadIdx.x >> 6;\n"
       "  const size_t r = blockIdx.x * size_t(64) + lane;\n"
       "  const bool live = r < p.n;\n"
       "  E2 s = e2(0);\n"
       "  if (live) {\n"
    << "    const u64* __restrict__ a = p.args + r * " << aw << ";\n"
    << "    const u64* __restrict__ m = p.mult + r * " << L << ";\n"
    << "    E2* __restrict__ trow = p.terms + r * " << L << ";\n"
       "    switch (grp) {\n";
  for (unsigned g = 0; g < G; g++) {
    const size_t j0 = size_t(g) * 16, cnt = std::min<size_t>(16, L - j0);
    o << "    case " << g << ": {\n      E2 msg[16];\n";
    for (size_t t = 0; t < cnt; t++) {
      const size_t j = j0 + t;
      o << "      { GlAcc g0, g1; acc_init(g0); acc_init(g1);\n";
      for (size_t k = 0; k < counts[j]; k++)
        o << "        { const u64 v = a[" << offs[j] + k << "]; acc_mad(g0, v, p.ch->gp.g[" << k << "].c0); acc_mad(g1, v, p.ch->gp.g[" << k << "].c1); }\n";
      o << "        msg[" << t << "] = e2(gl_add(acc_reduce(g0), p.ch->beta.c0), gl_add(acc_reduce(g1), p.ch->beta.c1)); }\n";
    }
    o << "      e2_batch_inverse<16>(msg, " << cnt << ");\n";
    for (size_t t = 0; t < cnt; t++) {
      const size_t j = j0 + t;
      o << "      { const E2 v = e2_mul_base(msg[" << t << "], m[" << j << "]); trow[" << j << "] = v; s = e2_add(s, v); }\n";
    }
    o << "    } break;\n";
  }
  o << "    default: break;\n    }\n  }\n"
       "  part[grp][lane] = s;\n  __syncthreads();\n"
       "  if (grp == 0 && live) {\n    E2 t = part[0][lane];\n";
  for (unsigned g = 1; g < G; g++) o << "    t = e2_add(t, part[" << g << "][lane]);\n";
  o << "    p.rowsum[r] = t;\n  }\n}\n";
  return o.str();
}

std::string stage2_source(const std::vector<uint32_t>& counts) {
  const size_t L = counts.size();
  size_t aw = 0;
  for (auto c : counts) aw += c;
  std::ostringstream o;
  o << "#include \"lookup_params.h\"\nusing namespace msamd;\n"
       "extern \"C\" __global__ __launch_bounds__(256) void stage2_terms_jit(Stage2Params p) {\n"
       "  const size_t r = blockIdx.x * size_t(blockDim.x) + threadIdx.x;\n"
       "  if (r >= p.n) return;\n"
    << "  const u64* __restrict__ a = p.args + r * " << aw << ";\n"
    << "  const u64* __restrict__ m = p.mult + r * " << L << ";\n";
  if (aw % 2 == 0) {
    for (size_t k = 0; k < aw; k += 2)
      o << "  const ulonglong2 q" << k << " = reinterpret_cast<const ulonglong2*>(a)[" << k / 2 << "]; const u64 a" << k << " = q" << k
        << ".x, a" << k + 1 << " = q" << k << ".y;\n";
  } else {
    for (size_t k = 0; k < aw; k++) o << "  const u64 a" << k << " = a[" << k << "];\n";
  }
  for (size_t j = 0; j < L; j++) o << "  const u64 m" << j << " = m[" << j << "];\n";
  o << "  E2 s = e2(0);\n  E2* __restrict__ trow = p.terms + r * " << L << ";\n";
  size_t off = 0;
  std::vector<size_t> offs;
  for (auto c : counts) {
    offs.push_back(off);
    off += c;
  }
  for (size_t j0 = 0; j0 < L; j0 += 16) {
    const size_t cnt = std::min<size_t>(16, L - j0);
    o << "  {\n    E2 msg[16];\n";
    for (size_t t = 0; t < cnt; t++) {
      const size_t j = j0 + t;
      o << "    { GlAcc g0, g1; acc_init(g0); acc_init(g1);\n";
      for (size_t k = 0; k < counts[j]; k++)
        o << "      acc_mad(g0, a" << offs[j] + k << ", p.ch->gp.g[" << k << "].c0); acc_mad(g1, a" << offs[j] + k << ", p.ch->gp.g[" << k << "].c1);\n";
      o << "      msg[" << t << "] = e2(gl_add(acc_reduce(g0), p.ch->beta.c0), gl_add(acc_reduce(g1), p.ch->beta.c1)); }\n";
    }
    o << "    e2_batch_inverse<16>(msg, " << cnt << ");\n";
    for (size_t t = 0; t < cnt; t++) {
      const size_t j = j0 + t;
      o << "    { const E2 v = e2_mul_base(msg[" << t << "], m" << j << "); trow[" << j << "] = v; s = e2_add(s, v); }\n";
    }
    o << "  }\n";
  }
  o << "  p.rowsum[r] = s;\n}\n";
  return o.str();
}
// The terms pass with the lookup expressions evaluated in the kernel from the row-major trace (this row and the next one,
// wrapping): the lookup prefix of the node program as straight-line code, then the same fingerprints and batch inversion.
std::string stage2_trace_source(const std::vector<PNode>& nodes, const std::vector<std::pair<uint32_t, std::vector<uint32_t>>>& lookups,
                                size_t main_w, size_t pre_w) {
  const size_t nn = nodes.size(), L = lookups.size();
  std::vector<char> needed(nn, 0);
  for (auto& l : lookups) {
    needed[l.first] = 1;
    for (auto a : l.second) needed[a] = 1;
  }
  for (size_t i = nn; i-- > 0;) {
    if (!needed[i]) continue;
    const PNode& n = nodes[i];
    if (n.kind == OP_ADD || n.kind == OP_SUB || n.kind == OP_MUL) needed[n.a] = needed[n.b] = 1;
    if (n.kind == OP_NEG) needed[n.a] = 1;
  }
  // which columns of which row are read: [source 0 = preprocessed, 1 = main][offset][column]
  std::vector<char> used[2][2];
  for (int s2 = 0; s2 < 2; s2++)
    for (int of = 0; of < 2; of++) used[s2][of].assign(s2 ? main_w : pre_w, 0);
  for (size_t i = 0; i < nn; i++)
    if (needed[i] && nodes[i].kind == OP_VAR) used[nodes[i].source][nodes[i].offset][nodes[i].a] = 1;
  std::ostringstream o;
  o << "#include \"lookup_params.h\"\nusing namespace msamd;\n"
       "extern \"C\" __global__ __launch_bounds__(256) void stage2_terms_trace_jit(Stage2TraceParams p) {\n"
       "  const size_t r = blockIdx.x * size_t(blockDim.x) + threadIdx.x;\n"
       "  if (r >= p.n) return;\n"
       "  const size_t rn = r + 1 == p.n ? 0 : r + 1;\n"
       "  const u64 is_first = r == 0, is_last = r + 1 == p.n, is_trans = r + 1 != p.n;\n"
       "  (void)is_first; (void)is_last; (void)is_trans; (void)rn;\n";
  for (int s2 = 0; s2 < 2; s2++)
    for (int of = 0; of < 2; of++) {
      const size_t W = s2 ? main_w : pre_w;
      const char* base = s2 ? "p.trace" : "p.pre";
      const std::string nm = std::string(s2 ? "m" : "q") + (of ? "n" : "c");  // mc / mn / qc / qn + column
      bool any = false;
      for (char u : used[s2][of]) any = any || u;
      if (!any) continue;
      o << "  const u64* __restrict__ " << nm << " = " << base << " + " << (of ? "rn" : "r") << " * " << W << ";\n";
      if (W % 2 == 0) {  // rows are 16-byte aligned: fetch pairs with one load
        for (size_t k = 0; k < W; k += 2)
          if (used[s2][of][k] || used[s2][of][k + 1])
            o << "  const ulonglong2 " << nm << "p" << k << " = reinterpret_cast<const ulonglong2*>(" << nm << ")[" << k / 2 << "]; const u64 "
              << nm << k << " = " << nm << "p" << k << ".x, " << nm << k + 1 << " = " << nm << "p" << k << ".y; (void)" << nm << k << "; (void)"
              << nm << k + 1 << ";\n";
      } else {
        for (size_t k = 0; k < W; k++)
          if (used[s2][of][k]) o << "  const u64 " << nm << k << " = " << nm << "[" << k << "];\n";
      }
    }
  for (size_t i = 0; i < nn; i++) {
    if (!needed[i]) continue;
    const PNode& n = nodes[i];
    o << "  const u64 v" << i << " = ";
    switch (n.kind) {
      case OP_CONST: o << n.a << "ULL"; break;
      case OP_VAR: o << (n.source ? "m" : "q") << (n.offset ? "n" : "c") << n.a; break;
      case OP_IS_FIRST: o << "is_first"; break;
      case OP_IS_LAST: o << "is_last"; break;
      case OP_IS_TRANS: o << "is_trans"; break;
      case OP_ADD: o << "gl_add(v" << n.a << ", v" << n.b << ")"; break;
      case OP_SUB: o << "gl_sub(v" << n.a << ", v" << n.b << ")"; break;
      case OP_MUL: o << "gl_mul(v" << n.a << ", v" << n.b << ")"; break;
      case OP_NEG: o << "gl_neg(v" << n.a << ")"; break;
      default: o << "0ULL"; break;  // publics / stage-2 columns cannot occur here (checked by the caller)
    }
    o << ";\n";
  }
  o << "  E2 s = e2(0);\n  E2* __restrict__ trow = p.terms + r * " << L << ";\n";
  for (size_t j0 = 0; j0 < L; j0 += 16) {
    const size_t cnt = std::min<size_t>(16, L - j0);
    o << "  {\n    E2 msg[16];\n";
    for (size_t t = 0; t < cnt; t++) {
      const auto& l = lookups[j0 + t];
      o << "    { GlAcc g0, g1; acc_init(g0); acc_init(g1);\n";
      for (size_t k = 0; k < l.second.size(); k++)
        o << "      acc_mad(g0, v" << l.second[k] << ", p.ch->gp.g[" << k << "].c0); acc_mad(g1, v" << l.second[k] << ", p.ch->gp.g[" << k << "].c1);\n";
      o << "      msg[" << t << "] = e2(gl_add(acc_reduce(g0), p.ch->beta.c0), gl_add(acc_reduce(g1), p.ch->beta.c1)); }\n";
    }
    o << "    e2_batch_inverse<16>(msg, " << cnt << ");\n";
    for (size_t t = 0; t < cnt; t++) {
      const size_t j = j0 + t;
      o << "    { const E2 v = e2_mul_base(msg[" << t << "], v" << lookups[j].first << "); trow[" << j << "] = v; s = e2_add(s, v); }\n";
    }
    o << "  }\n";
  }
  o << "  p.rowsum[r] = s;\n}\n";
  return o.str();
}
}  // namespace

// Compiles (or fetches) the circuit's kernel; leaves `out` empty when the interpreter should be used.
void quotient_jit_build(const std::vector<PNode>& nodes, const std::vector<uint32_t>& zeros,
                        const std::vector<std::pair<uint32_t, std::vector<uint32_t>>>& lookups, size_t quotient_degree, JitKernel& out) {
  if (getenv("MSAMD_NO_JIT")) return;
  if (nodes.size() > 3000) return;  // ~15 s of hiprtc and 200 KB of straight-line code: the interpreter is the better deal
  const bool inl = inline_tables_fit(zeros.size(), lookups.size(), quotient_degree);
  load_kernel(code_object(circuit_source(nodes, zeros, lookups, inl)), "quotient_jit", out);
  out.inline_tables = out.function && inl;
}

void bb_quotient_jit_build(const std::vector<PNode>& nodes, const std::vector<uint32_t>& zeros,
                           const std::vector<std::pair<uint32_t, std::vector<uint32_t>>>& lookups, JitKernel& out) {
  if (getenv("MSAMD_NO_JIT") || nodes.size() > 3000) return;
  load_kernel(code_object(bb_circuit_source(nodes, zeros, lookups)), "bb_quotient_jit", out);
}

void bb_quotient_jit_launch(Ctx& ctx, const JitKernel& k, const void* args, size_t args_size, size_t rows) {
  std::vector<char> copy((const char*)args, (const char*)args + args_size);
  size_t size = args_size;
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, copy.data(), HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)k.function, (unsigned)((rows + 255) / 256), 1, 1, 256, 1, 1, 0, ctx.stream, nullptr, config));
}

void stage2_jit_build(const std::vector<uint32_t>& arg_counts, JitKernel& out) {
  if (getenv("MSAMD_NO_JIT") || arg_counts.empty() || arg_counts.size() > 256) return;
  size_t aw = 0;
  for (auto c : arg_counts) {
    if (c > (uint32_t)MAX_GPOW) return;  // the generic kernel's Horner path handles very long argument lists
    aw += c;
  }
  if (aw > 1024) return;
  const unsigned G = stage2_groups(arg_counts.size());
  load_kernel(code_object(G ? stage2_grouped_source(arg_counts, G) : stage2_source(arg_counts)), "stage2_terms_jit", out);
  out.groups = out.function ? G : 0u;
}

void stage2_trace_jit_build(const std::vector<PNode>& nodes, const std::vector<std::pair<uint32_t, std::vector<uint32_t>>>& lookups,
                            size_t main_w, size_t pre_w, size_t prefix_len, JitKernel& out) {
  if (getenv("MSAMD_NO_JIT") || getenv("MSAMD_NO_STAGE2_FUSION") || lookups.empty() || lookups.size() > 256 || prefix_len > 3000) return;
  size_t aw = 0;
  for (auto& l : lookups) {
    if (l.second.size() > (size_t)MAX_GPOW) return;
    aw += l.second.size();
  }
  if (aw > 1024) return;
  for (size_t i = 0; i < prefix_len; i++)  // only trace columns and row selectors may feed a lookup at witness time
    if (nodes[i].kind == OP_PUBLIC || (nodes[i].kind == OP_VAR && nodes[i].source == 2)) return;
  load_kernel(code_object(stage2_trace_source(nodes, lookups, main_w, pre_w)), "stage2_terms_trace_jit", out);
}

void stage2_trace_jit_launch(Ctx& ctx, const JitKernel& k, const Stage2TraceParams& p) {
  Stage2TraceParams copy = p;
  size_t size = sizeof(Stage2TraceParams);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &copy, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)k.function, (unsigned)((p.n + 255) / 256), 1, 1, 256, 1, 1, 0, ctx.stream, nullptr, config));
}

void stage2_jit_launch(Ctx& ctx, const JitKernel& k, const Stage2Params& p) {
  Stage2Params copy = p;
  size_t size = sizeof(Stage2Params);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &copy, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  if (k.groups)  // 64 rows x `groups` waves per workgroup
    HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)k.function, (unsigned)((p.n + 63) / 64), 1, 1, 64 * k.groups, 1, 1, 0, ctx.stream, nullptr, config));
  else
    HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)k.function, (unsigned)((p.n + 255) / 256), 1, 1, 256, 1, 1, 0, ctx.stream, nullptr, config));
}

void quotient_jit_launch(Ctx& ctx, const JitKernel& k, const QParams& p, size_t nq) {
  QParams copy = p;
  size_t size = sizeof(QParams);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &copy, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)k.function, (unsigned)((nq + 255) / 256), 1, 1, 256, 1, 1, 0, ctx.stream, nullptr, config));
}

}  // namespace msamd
