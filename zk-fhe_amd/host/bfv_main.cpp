// `bfv` -- command-line driver with the reference's surface (README.md:18-52; clap `Cli` of halo2-scaffold,
// examples/bfv.rs:306-312):   bfv --name bfv -k 13 --input bfv/bfv.in {mock|keygen|prove|verify}
// Files: reads data/<input>, configs/<name>.json (prove); writes configs/<name>.json (keygen), data/<name>.snark (prove).
// keygen also writes data/<name>.vk and data/<name>.pk; `prove` loads the pk, `verify` reads vk + snark on the host CPU.
// SRS: params/kzg_bn254_<k>.srs (README.md:34, .gitignore:17; directory from $PARAMS_DIR as in halo2-base's gen_srs), halo2's
// ParamsKZG RawBytes layout.  keygen / prove read it when it exists and otherwise derive the reference's unsafe test setup
// (ChaCha20Rng::from_seed([0; 32]), zkfhe.h ZKFHE_SRS_HALO2_UNSAFE) and write it; verify reads only its G2 tail.
// --transcript poseidon|blake2b (keygen; default poseidon = the reference's PoseidonTranscript) is recorded in pk and vk.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include <sys/random.h>
#include <sys/stat.h>

#include "../../include/zkfhe.h"

static std::string slurp(const std::string &path) {
  std::ifstream f(path);
  if (!f) {
    fprintf(stderr, "cannot open %s\n", path.c_str());
    exit(2);
  }
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

// tiny reader for the pinning file: numbers after a key, and number arrays
static std::vector<uint32_t> numbers_after(const std::string &text, const std::string &key, size_t from = 0, size_t *end = nullptr) {
  std::vector<uint32_t> out;
  size_t p = text.find("\"" + key + "\"", from);
  if (p == std::string::npos) return out;
  p = text.find(':', p);
  size_t i = p + 1;
  while (i < text.size() && isspace((unsigned char)text[i])) ++i;
  bool arr = text[i] == '[';
  int depth = 0;
  for (; i < text.size(); ++i) {
    char c = text[i];
    if (c == '[') ++depth;
    else if (c == ']') {
      if (--depth == 0) { ++i; break; }
    } else if (isdigit((unsigned char)c)) {
      uint64_t v = 0;
      while (i < text.size() && isdigit((unsigned char)text[i])) v = v * 10 + (text[i++] - '0');
      out.push_back((uint32_t)v);
      --i;
      if (!arr) { ++i; break; }
    } else if (!arr && (c == ',' || c == '}')) break;
  }
  if (end) *end = i;
  return out;
}

struct Pinning {
  zkfhe_bfv_config c{};
  std::vector<uint32_t> bp0, bp1, bpr;
};

static bool load_pinning(const std::string &path, Pinning &p) {
  std::ifstream f(path);
  if (!f) return false;
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string t = ss.str();
  p.c.k = numbers_after(t, "degree")[0];
  p.c.n_rlc = numbers_after(t, "num_rlc_columns")[0];
  auto ra = numbers_after(t, "num_range_advice"), la = numbers_after(t, "num_lookup_advice");
  p.c.n_gate0 = ra[0];
  p.c.n_gate1 = ra[1];
  p.c.n_lookup = la[1];
  p.c.unusable_rows = numbers_after(t, "unusable_rows")[0];
  p.c.lookup_bits = numbers_after(t, "lookup_bits")[0];
  // "gate": [[..],[..],[]], "rlc": [..]
  size_t g = t.find("\"gate\"");
  size_t a = t.find('[', g), b1 = t.find('[', a + 1), e1 = t.find(']', b1), b2 = t.find('[', e1), e2 = t.find(']', b2);
  auto parse = [&](size_t lo, size_t hi) {
    std::vector<uint32_t> v;
    for (size_t i = lo; i < hi; ++i)
      if (isdigit((unsigned char)t[i])) {
        uint64_t x = 0;
        while (i < hi && isdigit((unsigned char)t[i])) x = x * 10 + (t[i++] - '0');
        v.push_back((uint32_t)x);
      }
    return v;
  };
  p.bp0 = parse(b1, e1);
  p.bp1 = parse(b2, e2);
  p.bpr = numbers_after(t, "rlc", e2);
  p.c.bp_gate0 = p.bp0.data();
  p.c.n_bp_gate0 = (uint32_t)p.bp0.size();
  p.c.bp_gate1 = p.bp1.data();
  p.c.n_bp_gate1 = (uint32_t)p.bp1.size();
  p.c.bp_rlc = p.bpr.data();
  p.c.n_bp_rlc = (uint32_t)p.bpr.size();
  p.c.replay = 1;
  return true;
}

static void write_pinning(const std::string &path, const zkfhe_bfv_config &c, const std::vector<uint32_t> bp[3]) {
  std::ofstream f(path);
  auto arr = [&](const std::vector<uint32_t> &v) {
    std::string s = "[";
    for (size_t i = 0; i < v.size(); ++i) s += (i ? "," : "") + std::to_string(v[i]);
    return s + "]";
  };
  f << "{\n  \"params\": {\n    \"degree\": " << c.k << ",\n    \"num_rlc_columns\": " << c.n_rlc << ",\n    \"num_range_advice\": [" << c.n_gate0 << ","
    << c.n_gate1 << ",0],\n    \"num_lookup_advice\": [0," << c.n_lookup << ",0],\n    \"num_fixed\": 1,\n    \"unusable_rows\": " << c.unusable_rows
    << ",\n    \"keccak_rows_per_round\": 50,\n    \"lookup_bits\": " << c.lookup_bits << "\n  },\n  \"break_points\": {\n    \"gate\": [" << arr(bp[0])
    << "," << arr(bp[1]) << ",[]],\n    \"rlc\": " << arr(bp[2]) << "\n  }\n}\n";
}

#define CHECK(x)                                                          \
  do {                                                                    \
    int rc_ = (x);                                                        \
    if (rc_) {                                                            \
      fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, zkfhe_last_error(ctx)); \
      return 1;                                                           \
    }                                                                     \
  } while (0)

int main(int argc, char **argv) {
  std::string name = "bfv", input, cmd, config_path = "configs", data_path = "data";
  unsigned k = 13;
  uint32_t transcript = ZKFHE_TRANSCRIPT_POSEIDON;
  zkfhe_bfv_params prm = {1024, 536870909ULL, 7, 19};  // examples/bfv.rs:27-30
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() { return std::string(i + 1 < argc ? argv[++i] : ""); };
    if (a == "--name" || a == "-n") name = next();
    else if (a == "-k" || a == "--degree") k = (unsigned)atoi(next().c_str());
    else if (a == "--input" || a == "-i") input = next();
    else if (a == "--config-path" || a == "-c") config_path = next();
    else if (a == "--data-path" || a == "-d") data_path = next();
    else if (a == "--ring-degree") prm.n = strtoull(next().c_str(), nullptr, 10);
    else if (a == "--modulus") prm.q = strtoull(next().c_str(), nullptr, 10);
    else if (a == "--transcript") {
      const std::string t = next();
      if (t == "poseidon") transcript = ZKFHE_TRANSCRIPT_POSEIDON;
      else if (t == "blake2b") transcript = ZKFHE_TRANSCRIPT_BLAKE2B;
      else {
        fprintf(stderr, "--transcript must be poseidon or blake2b\n");
        return 2;
      }
    }
    else if (a == "mock" || a == "keygen" || a == "prove" || a == "verify") cmd = a;
    else if (a == "--") continue;
    else {
      fprintf(stderr, "usage: bfv --name <n> -k <degree> --input <file under data/> {mock|keygen|prove|verify}\n");
      return 2;
    }
  }
  if (input.empty()) input = name + ".in";
  if (cmd.empty()) {
    fprintf(stderr, "missing command: mock | keygen | prove | verify\n");
    return 2;
  }
  const std::string text = slurp(data_path + "/" + input);
  const std::string pin_path = config_path + "/" + name + ".json";
  Pinning pin;
  const bool have_pin = load_pinning(pin_path, pin);
  pin.c.transcript = transcript;
  if (cmd == "mock") {
    // MockProver::run(..).assert_satisfied(): the circuit's own asserts while the witness is generated, then every gate,
    // lookup and copy constraint on every row of the assigned table
    if (!have_pin) {
      fprintf(stderr, "mock needs %s (run keygen first)\n", pin_path.c_str());
      return 1;
    }
    uint8_t gamma[32] = {7};
    char err[512] = {0};
    zkfhe_bfv_tables *t = nullptr;
    zkfhe_bfv_config c = pin.c;
    c.replay = 0;
    int rc = zkfhe_bfv_build_tables(text.c_str(), &prm, &c, gamma, 1, &t, err, sizeof(err));
    if (rc) {
      fprintf(stderr, "circuit is not satisfied: %s\n", err);
      return 1;
    }
    uint64_t failures = 0;
    rc = zkfhe_bfv_mock_check(t, gamma, &failures, err, sizeof(err));
    const size_t n_adv = zkfhe_bfv_tables_count(t, 0), n_rows = zkfhe_bfv_tables_count(t, 2), n_copies = zkfhe_bfv_tables_count(t, 4);
    zkfhe_bfv_tables_free(t);
    if (rc || failures) {
      fprintf(stderr, "Mock prover: circuit is NOT satisfied: %llu violated constraint(s); first: %s\n", (unsigned long long)failures, err);
      return 1;
    }
    printf("Mock prover: %zu advice columns x %zu rows, %zu copy constraints: every gate, lookup and copy constraint holds\n", n_adv, n_rows, n_copies);
    return 0;
  }
  const char *seed = ZKFHE_SRS_HALO2_UNSAFE;
  const char *pdir = getenv("PARAMS_DIR");
  const std::string params_dir = pdir && pdir[0] ? pdir : "params";
  const std::string srs_path = params_dir + "/kzg_bn254_" + std::to_string(k) + ".srs";
  if (cmd == "verify") {
    // README.md:48-52: reads data/<name>.vk and data/<name>.snark (host CPU only, no GPU needed)
    const std::string vk = slurp(data_path + "/" + name + ".vk"), sn = slurp(data_path + "/" + name + ".snark");
    std::vector<uint8_t> instv, proofv;
    if (zkfhe_snark_decode((const uint8_t *)sn.data(), sn.size(), nullptr, nullptr, nullptr, nullptr) != ZKFHE_OK) {
      fprintf(stderr, "%s/%s.snark is not a zkfhe snark file (or is truncated)\n", data_path.c_str(), name.c_str());
      return 1;
    }
    size_t ninst = 0, proof_len = 0;
    zkfhe_snark_decode((const uint8_t *)sn.data(), sn.size(), nullptr, &ninst, nullptr, &proof_len);
    instv.resize(32 * ninst + 1), proofv.resize(proof_len + 1);
    zkfhe_snark_decode((const uint8_t *)sn.data(), sn.size(), instv.data(), &ninst, proofv.data(), &proof_len);
    const uint8_t *inst = instv.data(), *proof = proofv.data();
    int ok = 0;
    char err[256] = {0};
    auto t0 = std::chrono::steady_clock::now();
    uint8_t g2[128], sg2[128];
    uint32_t fk = 0;
    if (zkfhe_srs_file_g2(srs_path.c_str(), &fk, g2, sg2) == ZKFHE_OK && fk == k) {
      zkfhe_bfv_verify_g2((const uint8_t *)vk.data(), vk.size(), inst, ninst, proof, proof_len, g2, sg2, &ok, err, sizeof(err));
    } else {   // no params file: the reference's unsafe setup, derived (gen_srs would do the same)
      zkfhe_bfv_verify((const uint8_t *)vk.data(), vk.size(), inst, ninst, proof, proof_len, (const uint8_t *)seed, strlen(seed), &ok, err, sizeof(err));
    }
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (!ok) {
      fprintf(stderr, "Snark verification FAILED%s%s\n", err[0] ? ": " : "", err);
      return 1;
    }
    printf("Snark verified successfully in %.3fms\n", ms);
    return 0;
  }
  setenv("ZKFHE_SPIN_WAIT", "1", 0);  // one proof at a time: spinning waits are ~1.4 ms faster per proof than sleeping ones
  // one command = one keygen or one proof: the 189 GB of digit-multiple tables a proving service builds once (2.7 s) would cost a
  // hundred proofs' time here; a 2 GB budget builds in ~30 ms and serves the narrow commitments
  setenv("ZKFHE_TABLE_GB", "2", 0);
  zkfhe_ctx *ctx = nullptr;
  if (zkfhe_ctx_create(0, nullptr, &ctx)) {
    fprintf(stderr, "no gfx950 device: %s\n", zkfhe_last_error(nullptr));
    return 1;
  }
  zkfhe_srs *srs = nullptr;
  {
    struct stat sb;
    if (stat(srs_path.c_str(), &sb) == 0) {
      CHECK(zkfhe_srs_load(ctx, srs_path.c_str(), &srs));
    } else {
      CHECK(zkfhe_srs_create(ctx, k, (const uint8_t *)seed, strlen(seed), &srs));
      mkdir(params_dir.c_str(), 0755);
      CHECK(zkfhe_srs_save(ctx, srs, srs_path.c_str()));
      printf("wrote %s (unsafe test setup: ChaCha20Rng::from_seed([0; 32]))\n", srs_path.c_str());
      zkfhe_srs_drop_host_copy(srs);   // written: the host copies of the points are not needed again
    }
  }
  zkfhe_bfv_pk *pk = nullptr;
  if (cmd == "keygen") {
    zkfhe_bfv_config c{};
    if (have_pin) c = pin.c;
    else {
      // no pinning yet: halo2-base auto-configuration, as the reference's keygen does (README.md:28-38)
      uint32_t counts[4];
      char aerr[256] = {0};
      if (zkfhe_bfv_auto_config(text.c_str(), &prm, k, 109, 8, counts, aerr, sizeof(aerr))) {
        fprintf(stderr, "keygen: auto-configuration failed: %s\n", aerr);
        return 1;
      }
      c.n_gate0 = counts[0], c.n_gate1 = counts[1], c.n_lookup = counts[2], c.n_rlc = counts[3];
      c.unusable_rows = 109, c.lookup_bits = 8;
      c.transcript = transcript;
      printf("auto-configured columns: gate %u + %u, lookup %u, rlc %u\n", counts[0], counts[1], counts[2], counts[3]);
    }
    c.replay = 0;
    c.k = k;
    CHECK(zkfhe_bfv_keygen(ctx, srs, text.c_str(), &prm, &c, &pk));
    std::vector<uint32_t> bp[3];
    for (int w = 0; w < 3; ++w) {
      uint32_t cnt = 0;
      zkfhe_bfv_pk_break_points(pk, w, nullptr, &cnt);
      bp[w].resize(cnt);
      zkfhe_bfv_pk_break_points(pk, w, bp[w].data(), &cnt);
    }
    write_pinning(pin_path, c, bp);
    uint8_t d[32];
    zkfhe_bfv_pk_info(pk, d, nullptr, nullptr);
    size_t vlen = 0;
    zkfhe_bfv_pk_export_vk(pk, nullptr, 0, &vlen);
    std::vector<uint8_t> vkb(vlen);
    CHECK(zkfhe_bfv_pk_export_vk(pk, vkb.data(), vkb.size(), &vlen));
    std::ofstream(data_path + "/" + name + ".vk", std::ios::binary).write((const char *)vkb.data(), (std::streamsize)vlen);
    CHECK(zkfhe_bfv_pk_save(ctx, pk, (data_path + "/" + name + ".pk").c_str()));   // README.md:38: keygen writes data/<name>.pk
    printf("keygen done; pinning written to %s, keys to %s/%s.{pk,vk}; vk digest ", pin_path.c_str(), data_path.c_str(), name.c_str());
    for (int i = 31; i >= 0; --i) printf("%02x", d[i]);
    printf("\n");
  } else {
    if (!have_pin) {
      fprintf(stderr, "prove needs %s (run keygen first)\n", pin_path.c_str());
      return 1;
    }
    const std::string pk_path = data_path + "/" + name + ".pk";
    if (FILE *pf = fopen(pk_path.c_str(), "rb")) {
      fclose(pf);
      CHECK(zkfhe_bfv_pk_load(ctx, srs, pk_path.c_str(), &pk));   // the key written by keygen
    } else {
      // no key on disk: rebuild it; structure comes from an all-zero input of the same shape (bfv_empty.in, README.md:31)
      std::string empty = text;
      for (size_t i = 0; i + 1 < empty.size(); ++i)
        if (empty[i] == '"' && isdigit((unsigned char)empty[i + 1])) {
          size_t j = empty.find('"', i + 1);
          empty.replace(i + 1, j - i - 1, "0");
        }
      CHECK(zkfhe_bfv_keygen(ctx, srs, empty.c_str(), &prm, &pin.c, &pk));
    }
    std::vector<uint8_t> proof(1 << 20), instb((size_t)32 << 20);
    size_t len = 0, ninst = instb.size() / 32;
    // the blinding stream: zero knowledge rests on this seed being fresh and secret, so no fallback if the OS has none
    uint8_t seed32[32];
    if (getrandom(seed32, 32, 0) != 32) {
      fprintf(stderr, "prove: the operating system gave no randomness (getrandom): refusing to prove with a predictable blinding seed\n");
      return 1;
    }
    float tm[5];
    auto t0 = std::chrono::steady_clock::now();
    CHECK(zkfhe_bfv_prove(ctx, srs, pk, text.c_str(), seed32, proof.data(), proof.size(), &len, instb.data(), &ninst, tm));
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    {
      // data/<name>.snark: zkfhe.h zkfhe_snark_encode ("ZKFHESN2": the instances and proof fields of snark-verifier-sdk's Snark as bincode writes them)
      size_t slen = 0;
      zkfhe_snark_encode(instb.data(), ninst, proof.data(), len, nullptr, 0, &slen);
      std::vector<uint8_t> sb(slen);
      CHECK(zkfhe_snark_encode(instb.data(), ninst, proof.data(), len, sb.data(), sb.size(), &slen));
      std::ofstream f(data_path + "/" + name + ".snark", std::ios::binary);
      f.write((const char *)sb.data(), (std::streamsize)slen);
    }
    printf("Proving time: %.3fms  (witness %.1f, commit %.1f, quotient %.1f, open %.1f)\n", ms, tm[0], tm[1], tm[2], tm[3]);
    printf("proof: %zu bytes, %zu public inputs -> %s/%s.snark\n", len, ninst, data_path.c_str(), name.c_str());
  }
  zkfhe_bfv_pk_destroy(ctx, pk);
  zkfhe_srs_destroy(ctx, srs);
  zkfhe_ctx_destroy(ctx);
  return 0;
}
