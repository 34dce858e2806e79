// Command-line benchmark for the BLAS-3 entry points of libsbamd, taking the
// reference's benchmark parameter files unchanged.
//
//   bench_blas3 <gemm|gemm_batched|gemm_batched_strided|symm|trsm> [options]
//
// CSV row formats (benchmark/README.md:111-166 of the reference):
//   gemm                  transA,transB,m,k,n,alpha,beta
//   gemm_batched          transA,transB,m,k,n,alpha,beta,batch_size,batch_type
//   gemm_batched_strided  transA,transB,m,k,n,alpha,beta,batch_size,stride_a_mul,stride_b_mul,stride_c_mul
//   symm                  side,uplo,m,n,alpha,beta
//   trsm                  side,uplo,trans,diag,m,n,alpha
// Rows with complex scalars (two values each for alpha and beta) keep the real parts,
// as common/include/common/common_utils.hpp:503-514 does.  Without --csv-param the
// reference's default sweeps are used (common_utils.hpp:469-490, :725-744, :787-805,
// :933-948, :1015-1032).  Counters per row follow
// common/include/common/blas3_state_counters.hpp and common_utils.hpp:1893-1918:
// n_fl_ops, bytes_processed, avg/best event time and avg/best overall time in ns.
//
// Plain C++17: needs only include/sbamd_gemm.h and libsbamd.so.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <regex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "sbamd_gemm.h"

namespace {

struct Args {
  std::string op;
  std::string csv_param;
  std::string dtype = "float";
  std::string format = "console";
  std::string out_path;
  std::string out_format = "json";
  std::string filter;
  std::string kernel = "auto";
  double min_time = 0.25;  // seconds of measured calls per row
  int device = 0;
  bool list_only = false;
};

[[noreturn]] void usage(int code) {
  std::fprintf(code ? stderr : stdout,
               "usage: bench_blas3 <gemm|gemm_batched|gemm_batched_strided|symm|trsm> [options]\n"
               "  --csv-param=FILE             parameter rows in the reference's CSV format\n"
               "  --dtype=float|double|half|bf16   (half: half in/out; bf16: bf16 in, float out; gemm ops only)\n"
               "  --device=N                   HIP device ordinal (default 0)\n"
               "  --kernel=auto|large|small|huge|tiny|splitk|naive   pin a GEMM kernel family\n"
               "  --benchmark_format=console|json|csv\n"
               "  --benchmark_out=FILE  --benchmark_out_format=json|csv|console\n"
               "  --benchmark_filter=REGEX     run only rows whose name matches\n"
               "  --benchmark_min_time=SECONDS measured time per row (default 0.25)\n"
               "  --benchmark_list_tests[=true]  print the row names and exit\n");
  std::exit(code);
}

bool take(const std::string& a, const char* key, std::string* value) {
  const std::string k = std::string(key) + "=";
  if (a.compare(0, k.size(), k) != 0) return false;
  *value = a.substr(k.size());
  return true;
}

Args parse_args(int argc, char** argv) {
  Args args;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    std::string v;
    if (a == "--help" || a == "-h") usage(0);
    if (take(a, "--csv-param", &args.csv_param) || take(a, "--dtype", &args.dtype) ||
        take(a, "--benchmark_format", &args.format) || take(a, "--benchmark_out", &args.out_path) ||
        take(a, "--benchmark_out_format", &args.out_format) || take(a, "--benchmark_filter", &args.filter) ||
        take(a, "--kernel", &args.kernel))
      continue;
    if (take(a, "--benchmark_min_time", &v)) {
      args.min_time = std::stod(v);  // google-benchmark also accepts a trailing 's'
      continue;
    }
    if (take(a, "--device", &v)) {
      args.device = std::stoi(v);
      continue;
    }
    if (a == "--benchmark_list_tests" || a == "--benchmark_list_tests=true") {
      args.list_only = true;
      continue;
    }
    if (a == "--benchmark_list_tests=false") continue;
    if (!a.empty() && a[0] != '-' && args.op.empty()) {
      args.op = a;
      continue;
    }
    std::fprintf(stderr, "unknown argument: %s\n", a.c_str());
    usage(2);
  }
  if (args.op.empty()) usage(2);
  return args;
}

// ------------------------------------------------------------------ CSV rows
std::vector<std::vector<std::string>> read_csv(const std::string& path) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("could not open " + path);
  std::vector<std::vector<std::string>> rows;
  std::string line;
  while (std::getline(in, line)) {
    while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
    if (line.empty() || line[0] == '#') continue;
    std::vector<std::string> cells;
    std::stringstream ss(line);
    std::string cell;
    while (std::getline(ss, cell, ',')) {
      const size_t b = cell.find_first_not_of(" \t");
      const size_t e = cell.find_last_not_of(" \t");
      cells.push_back(b == std::string::npos ? "" : cell.substr(b, e - b + 1));
    }
    rows.push_back(cells);
  }
  return rows;
}

int64_t to_int(const std::string& s) {
  size_t pos = 0;
  const long long v = std::stoll(s, &pos);
  if (pos != s.size()) throw std::runtime_error("invalid parameter: " + s);
  return v;
}
double to_real(const std::string& s) {
  size_t pos = 0;
  const double v = std::stod(s, &pos);
  if (pos != s.size()) throw std::runtime_error("invalid parameter: " + s);
  return v;
}
char to_flag(const std::string& s) {
  if (s.empty()) throw std::runtime_error("invalid parameter: empty flag");
  return char(std::tolower(static_cast<unsigned char>(s[0])));
}

struct Row {
  // gemm family
  char ta = 'n', tb = 'n';
  int64_t m = 0, k = 0, n = 0;
  double alpha = 1, beta = 0;
  int64_t batch = 1;
  int batch_type = SBAMD_BATCH_STRIDED;
  int64_t sa_mul = 1, sb_mul = 1, sc_mul = 1;
  // symm / trsm
  char side = 'l', uplo = 'l', trans = 'n', diag = 'n';
};

std::vector<Row> default_rows(const std::string& op) {
  std::vector<Row> rows;
  const char tr[2] = {'n', 't'};
  if (op == "gemm") {
    for (char t1 : tr)
      for (char t2 : tr)
        for (int64_t m = 32; m <= 8192; m *= 8)
          for (int64_t k = 32; k <= 8192; k *= 8)
            for (int64_t n = 32; n <= 8192; n *= 8) {
              Row r;
              r.ta = t1, r.tb = t2, r.m = m, r.k = k, r.n = n, r.alpha = 1, r.beta = 1;
              rows.push_back(r);
            }
  } else if (op == "gemm_batched" || op == "gemm_batched_strided") {
    for (char t1 : tr)
      for (char t2 : tr)
        for (int64_t s = 128; s <= 8192; s *= 8) {
          Row r;
          r.ta = t1, r.tb = t2, r.m = r.k = r.n = s, r.alpha = 1, r.beta = 1, r.batch = 8;
          if (op == "gemm_batched_strided") r.sa_mul = r.sb_mul = r.sc_mul = 2;
          rows.push_back(r);
        }
  } else if (op == "symm") {
    for (char side : {'l', 'r'})
      for (char uplo : {'u', 'l'})
        for (int64_t m = 32; m <= 8192; m *= 4) {
          Row r;
          r.side = side, r.uplo = uplo, r.m = r.n = m, r.alpha = 1, r.beta = 1;
          rows.push_back(r);
        }
  } else if (op == "trsm") {
    for (char side : {'l', 'r'})
      for (char uplo : {'u', 'l'})
        for (char trans : {'n', 't'})
          for (char diag : {'u', 'n'})
            for (int64_t m = 512; m <= 8192; m *= 4) {
              Row r;
              r.side = side, r.uplo = uplo, r.trans = trans, r.diag = diag, r.m = r.n = m, r.alpha = 1;
              rows.push_back(r);
            }
  }
  return rows;
}

int parse_batch_type(const std::string& s) {
  if (s == "strided" || s == "0") return SBAMD_BATCH_STRIDED;
  if (s == "interleaved" || s == "1") return SBAMD_BATCH_INTERLEAVED;
  throw std::runtime_error("invalid batch_type: " + s);
}

std::vector<Row> rows_from_csv(const std::string& op, const std::string& path) {
  std::vector<Row> rows;
  for (const auto& v : read_csv(path)) {
    Row r;
    const size_t c = v.size();
    if (op == "gemm" || op == "gemm_batched" || op == "gemm_batched_strided") {
      const size_t base = op == "gemm" ? 7 : op == "gemm_batched" ? 9 : 11;
      if (c != base && c != base + 2)
        throw std::runtime_error("invalid number of parameters (" + std::to_string(base) + " expected)");
      const bool cplx = c == base + 2;
      r.ta = to_flag(v[0]), r.tb = to_flag(v[1]);
      r.m = to_int(v[2]), r.k = to_int(v[3]), r.n = to_int(v[4]);
      r.alpha = to_real(v[5]);
      r.beta = to_real(v[cplx ? 7 : 6]);
      const size_t rest = cplx ? 9 : 7;
      if (op != "gemm") r.batch = to_int(v[rest]);
      if (op == "gemm_batched") r.batch_type = parse_batch_type(v[rest + 1]);
      if (op == "gemm_batched_strided")
        r.sa_mul = to_int(v[rest + 1]), r.sb_mul = to_int(v[rest + 2]), r.sc_mul = to_int(v[rest + 3]);
    } else if (op == "symm") {
      if (c != 6) throw std::runtime_error("invalid number of parameters (6 expected)");
      r.side = to_flag(v[0]), r.uplo = to_flag(v[1]), r.m = to_int(v[2]), r.n = to_int(v[3]);
      r.alpha = to_real(v[4]), r.beta = to_real(v[5]);
    } else {
      if (c != 7) throw std::runtime_error("invalid number of parameters (7 expected)");
      r.side = to_flag(v[0]), r.uplo = to_flag(v[1]), r.trans = to_flag(v[2]), r.diag = to_flag(v[3]);
      r.m = to_int(v[4]), r.n = to_int(v[5]), r.alpha = to_real(v[6]);
    }
    rows.push_back(r);
  }
  return rows;
}

// --------------------------------------------------------------------- types
struct DType {
  const char* name;   // as printed in the benchmark name
  int fill_code;      // sbamd_fill_uniform dtype of the inputs
  size_t in_bytes, out_bytes;
};

DType dtype_of(const std::string& s) {
  if (s == "float" || s == "f32") return {"float", 0, 4, 4};
  if (s == "double" || s == "f64") return {"double", 1, 8, 8};
  if (s == "half" || s == "f16") return {"half", 2, 2, 2};
  if (s == "bf16" || s == "bfloat16") return {"bfloat16", 3, 2, 4};
  throw std::runtime_error("unknown --dtype " + s);
}

void check(int st, sbamd_handle_t h, const char* what) {
  if (st == SBAMD_OK) return;
  std::string msg = std::string(what) + ": " + sbamd_status_string(st);
  if (h && st >= SBAMD_HIP_ERROR) msg += std::string(" (") + sbamd_last_error(h) + ")";
  throw std::runtime_error(msg);
}

struct DeviceBuf {
  sbamd_handle_t h = nullptr;
  void* p = nullptr;
  DeviceBuf(sbamd_handle_t handle, size_t bytes) : h(handle) {
    check(sbamd_malloc(h, &p, bytes ? bytes : 1), h, "sbamd_malloc");
  }
  ~DeviceBuf() {
    if (p) sbamd_free(h, p);
  }
  DeviceBuf(const DeviceBuf&) = delete;
  DeviceBuf& operator=(const DeviceBuf&) = delete;
};

// ------------------------------------------------------------------- results
struct Result {
  std::string name;
  int64_t iterations = 0;
  double n_fl_ops = 0, bytes_processed = 0;
  double total_event_ns = 0, best_event_ns = 0, total_overall_ns = 0, best_overall_ns = 0;
  std::string kernel;  // family the dispatcher picked (gemm ops)
  std::string error;
  double avg_event_ns() const { return iterations ? total_event_ns / iterations : 0; }
  double avg_overall_ns() const { return iterations ? total_overall_ns / iterations : 0; }
  double gflops() const { return avg_event_ns() > 0 ? n_fl_ops / avg_event_ns() : 0; }
  double gbytes() const { return avg_event_ns() > 0 ? bytes_processed / avg_event_ns() : 0; }
};

const char* kernel_name(int k) {
  switch (k) {
    case SBAMD_KERNEL_LARGE: return "large";
    case SBAMD_KERNEL_SMALL: return "small";
    case SBAMD_KERNEL_SPLITK: return "splitk";
    case SBAMD_KERNEL_NAIVE: return "naive";
    case SBAMD_KERNEL_HUGE: return "huge";
    case SBAMD_KERNEL_INTERLEAVED: return "interleaved";
    case SBAMD_KERNEL_TINY: return "tiny";
    case SBAMD_KERNEL_STREAM: return "stream";
    case SBAMD_KERNEL_MEDIUM: return "medium";
    default: return "-";
  }
}

int kernel_code(const std::string& s) {
  if (s == "auto") return SBAMD_KERNEL_AUTO;
  if (s == "large") return SBAMD_KERNEL_LARGE;
  if (s == "small") return SBAMD_KERNEL_SMALL;
  if (s == "huge") return SBAMD_KERNEL_HUGE;
  if (s == "splitk") return SBAMD_KERNEL_SPLITK;
  if (s == "naive") return SBAMD_KERNEL_NAIVE;
  if (s == "tiny") return SBAMD_KERNEL_TINY;
  if (s == "medium") return SBAMD_KERNEL_MEDIUM;
  throw std::runtime_error("unknown --kernel " + s);
}

// One warm-up call, then timed calls until min_time of device time has passed
// (at least 3, at most 10000): event time brackets the call on the stream, overall
// time is host wall-clock around call + wait, as blas_benchmark::utils::timef does.
void measure(sbamd_handle_t h, double min_time, const std::function<int()>& call, Result* r) {
  check(call(), h, r->name.c_str());
  check(sbamd_wait(h), h, "sbamd_wait");
  r->best_event_ns = r->best_overall_ns = 1e300;
  double spent = 0;
  while ((spent < min_time * 1e9 || r->iterations < 3) && r->iterations < 10000) {
    sbamd_event_t e0 = nullptr, e1 = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    check(sbamd_event_record(h, &e0), h, "event");
    check(call(), h, r->name.c_str());
    check(sbamd_event_record(h, &e1), h, "event");
    check(sbamd_wait(h), h, "sbamd_wait");
    const auto t1 = std::chrono::steady_clock::now();
    float ms = 0;
    check(sbamd_event_elapsed_ms(e0, e1, &ms), h, "elapsed");
    sbamd_event_destroy(e0);
    sbamd_event_destroy(e1);
    const double ev = double(ms) * 1e6;
    const double ov = std::chrono::duration<double, std::nano>(t1 - t0).count();
    r->total_event_ns += ev;
    r->total_overall_ns += ov;
    r->best_event_ns = std::min(r->best_event_ns, ev);
    r->best_overall_ns = std::min(r->best_overall_ns, ov);
    r->iterations += 1;
    spent += std::max(ev, ov);
  }
}

std::string fmt_scalar(double v) {
  std::ostringstream s;
  s << v;
  return s.str();
}

std::string row_name(const std::string& op, const DType& dt, const Row& r) {
  std::ostringstream s;
  auto head = [&](const char* opname) { s << "BM_" << opname << "<" << dt.name << ">/"; };
  if (op == "gemm") {
    head("Gemm");
    s << r.ta << "/" << r.tb << "/" << r.m << "/" << r.k << "/" << r.n;
  } else if (op == "gemm_batched") {
    head("GemmBatched");
    s << r.ta << "/" << r.tb << "/" << r.m << "/" << r.k << "/" << r.n << "/" << r.batch << "/"
      << (r.batch_type == SBAMD_BATCH_INTERLEAVED ? "interleaved" : "strided");
  } else if (op == "gemm_batched_strided") {
    head("GemmBatchedStrided");
    s << r.ta << "/" << r.tb << "/" << r.m << "/" << r.k << "/" << r.n << "/" << r.batch << "/" << r.sa_mul << "/"
      << r.sb_mul << "/" << r.sc_mul;
  } else if (op == "symm") {
    head("Symm");
    s << r.side << "/" << r.uplo << "/" << r.m << "/" << r.n << "/" << fmt_scalar(r.alpha) << "/" << fmt_scalar(r.beta);
  } else {
    head("Trsm");
    s << r.side << "/" << r.uplo << "/" << r.trans << "/" << r.diag << "/" << r.m << "/" << r.n << "/"
      << fmt_scalar(r.alpha);
  }
  s << "/usm";
  return s.str();
}

// --------------------------------------------------------------- the workers
int call_gemm(sbamd_handle_t h, const DType& dt, const Row& r, const void* a, int64_t lda, int64_t sa, const void* b,
              int64_t ldb, int64_t sb, void* c, int64_t ldc, int64_t sc, bool strided_call) {
  const std::string t = dt.name;
  if (r.batch_type == SBAMD_BATCH_INTERLEAVED && !strided_call) {
    if (t == "float")
      return sbamd_sgemm_batched(h, r.ta, r.tb, r.m, r.n, r.k, float(r.alpha), (const float*)a, lda, (const float*)b,
                                 ldb, float(r.beta), (float*)c, ldc, r.batch, SBAMD_BATCH_INTERLEAVED);
    if (t == "double")
      return sbamd_dgemm_batched(h, r.ta, r.tb, r.m, r.n, r.k, r.alpha, (const double*)a, lda, (const double*)b, ldb,
                                 r.beta, (double*)c, ldc, r.batch, SBAMD_BATCH_INTERLEAVED);
    return SBAMD_NOT_SUPPORTED;
  }
  if (t == "float")
    return sbamd_sgemm_strided_batched(h, r.ta, r.tb, r.m, r.n, r.k, float(r.alpha), (const float*)a, lda, sa,
                                       (const float*)b, ldb, sb, float(r.beta), (float*)c, ldc, sc, r.batch);
  if (t == "double")
    return sbamd_dgemm_strided_batched(h, r.ta, r.tb, r.m, r.n, r.k, r.alpha, (const double*)a, lda, sa,
                                       (const double*)b, ldb, sb, r.beta, (double*)c, ldc, sc, r.batch);
  if (t == "half")
    return sbamd_hgemm_strided_batched(h, r.ta, r.tb, r.m, r.n, r.k, float(r.alpha), (const uint16_t*)a, lda, sa,
                                       (const uint16_t*)b, ldb, sb, float(r.beta), (uint16_t*)c, ldc, sc, r.batch);
  return sbamd_bsgemm_strided_batched(h, r.ta, r.tb, r.m, r.n, r.k, float(r.alpha), (const uint16_t*)a, lda, sa,
                                      (const uint16_t*)b, ldb, sb, float(r.beta), (float*)c, ldc, sc, r.batch);
}

Result run_gemm_row(sbamd_handle_t h, const Args& args, const DType& dt, const Row& r) {
  Result res;
  res.name = row_name(args.op, dt, r);
  // leading dimensions are the tight ones (benchmark/portblas/blas3/gemm.cpp:47-49)
  const int64_t lda = r.ta == 'n' ? r.m : r.k, ldb = r.tb == 'n' ? r.k : r.n, ldc = r.m;
  const int64_t size_a = r.m * r.k, size_b = r.k * r.n, size_c = r.m * r.n;
  const int64_t sa = size_a * r.sa_mul, sb = size_b * r.sb_mul, sc = size_c * r.sc_mul;
  const bool strided_op = args.op == "gemm_batched_strided";
  const int64_t na = strided_op ? size_a + (r.batch - 1) * sa : size_a * r.batch;
  const int64_t nb = strided_op ? size_b + (r.batch - 1) * sb : size_b * r.batch;
  const int64_t nc = strided_op ? size_c + (r.batch - 1) * sc : size_c * r.batch;
  // counters: blas3_state_counters.hpp:62-75
  const double md = double(r.m), nd = double(r.n), kd = double(r.k), bd = double(r.batch);
  res.n_fl_ops = (2 * kd * md * nd + md * nd + (r.beta != 0 ? 2 * md * nd : 0)) * bd;
  res.bytes_processed = ((md * kd + kd * nd) * double(dt.in_bytes) + (md * nd + (r.beta != 0 ? md * nd : 0)) * double(dt.out_bytes)) * bd;

  DeviceBuf a(h, size_t(na) * dt.in_bytes), b(h, size_t(nb) * dt.in_bytes), c(h, size_t(nc) * dt.out_bytes);
  check(sbamd_fill_uniform(h, a.p, na, dt.fill_code, 1, -1.0, 1.0), h, "fill A");
  check(sbamd_fill_uniform(h, b.p, nb, dt.fill_code, 2, -1.0, 1.0), h, "fill B");
  check(sbamd_memset(h, c.p, 0, size_t(nc) * dt.out_bytes), h, "zero C");
  auto call = [&]() {
    return call_gemm(h, dt, r, a.p, lda, sa, b.p, ldb, sb, c.p, ldc, sc, strided_op);
  };
  measure(h, args.min_time, call, &res);
  res.kernel = kernel_name(sbamd_last_kernel(h));
  return res;
}

template <typename T>
void fill_triangular_host(std::vector<T>& a, int64_t k, int64_t lda, char uplo, char diag) {
  // Diagonally dominant triangle (the idea of fill_trsm_matrix, test/blas_test.hpp:185-214)
  // so that repeated solves stay finite.  The unused triangle stays zero.
  uint64_t s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() {
    s ^= s << 13, s ^= s >> 7, s ^= s << 17;
    return double(s >> 11) / double(1ull << 53) * 2.0 - 1.0;
  };
  const double d = diag == 'u' ? 1.0 : 4.0;
  for (int64_t i = 0; i < k; ++i)
    for (int64_t j = 0; j <= i; ++j) {
      const double v = i == j ? d : rnd() / double(k);
      if (uplo == 'l') a[size_t(i + j * lda)] = T(v);
      else a[size_t(j + i * lda)] = T(v);
    }
}

template <typename T>
Result run_symm_trsm_row(sbamd_handle_t h, const Args& args, const DType& dt, const Row& r) {
  Result res;
  res.name = row_name(args.op, dt, r);
  const bool is_symm = args.op == "symm";
  const int64_t k = r.side == 'l' ? r.m : r.n;
  const int64_t lda = k, ldb = r.m, ldc = r.m;
  const double md = double(r.m), nd = double(r.n);
  if (is_symm) {  // blas3_state_counters.hpp:152-164
    res.n_fl_ops = (r.side == 'l' ? 2 * md * md * nd : 2 * nd * nd * nd) + (r.beta != 0 ? 2 * md * nd : 0);
    const double read_a = r.side == 'l' ? md * (md + 1) / 2 : nd * (nd + 1) / 2;
    res.bytes_processed = ((r.beta != 0 ? 2 : 1) * md * nd + md * nd + read_a) * sizeof(T);
  } else {  // :240-252
    res.n_fl_ops = (r.side == 'l' ? 2 * md * (md + 1) / 2 * nd : 2 * nd * (nd + 1) / 2 * md) + md * nd;
    const double read_a = r.side == 'l' ? md * (md + 1) / 2 : md * nd;
    const double read_b = r.side == 'r' ? nd * (nd + 1) / 2 : md * nd;
    res.bytes_processed = (read_a + read_b + md * nd) * sizeof(T);
  }
  DeviceBuf a(h, size_t(k * lda) * sizeof(T)), b(h, size_t(r.m * r.n) * sizeof(T)), c(h, size_t(r.m * r.n) * sizeof(T));
  if (is_symm) {
    check(sbamd_fill_uniform(h, a.p, k * lda, dt.fill_code, 1, -1.0, 1.0), h, "fill A");
  } else {
    std::vector<T> host(size_t(k * lda), T(0));
    fill_triangular_host(host, k, lda, r.uplo, r.diag);
    check(sbamd_copy_to_device(h, a.p, host.data(), host.size() * sizeof(T)), h, "copy A");
  }
  check(sbamd_fill_uniform(h, b.p, r.m * r.n, dt.fill_code, 2, -1.0, 1.0), h, "fill B");
  check(sbamd_memset(h, c.p, 0, size_t(r.m * r.n) * sizeof(T)), h, "zero C");
  std::function<int()> call;
  if constexpr (std::is_same<T, float>::value) {
    if (is_symm)
      call = [&]() { return sbamd_ssymm(h, r.side, r.uplo, r.m, r.n, float(r.alpha), (const float*)a.p, lda, (const float*)b.p, ldb, float(r.beta), (float*)c.p, ldc); };
    else
      call = [&]() { return sbamd_strsm(h, r.side, r.uplo, r.trans, r.diag, r.m, r.n, float(r.alpha), (const float*)a.p, lda, (float*)b.p, ldb); };
  } else {
    if (is_symm)
      call = [&]() { return sbamd_dsymm(h, r.side, r.uplo, r.m, r.n, r.alpha, (const double*)a.p, lda, (const double*)b.p, ldb, r.beta, (double*)c.p, ldc); };
    else
      call = [&]() { return sbamd_dtrsm(h, r.side, r.uplo, r.trans, r.diag, r.m, r.n, r.alpha, (const double*)a.p, lda, (double*)b.p, ldb); };
  }
  measure(h, args.min_time, call, &res);
  return res;
}

// ------------------------------------------------------------------ printing
void print_console(std::ostream& os, const std::vector<Result>& results) {
  size_t w = 20;
  for (const auto& r : results) w = std::max(w, r.name.size());
  char line[512];
  std::snprintf(line, sizeof line, "%-*s %12s %12s %8s %11s %10s %8s\n", int(w), "Benchmark", "avg_event_us",
                "best_event_us", "iters", "GFLOP/s", "GB/s", "kernel");
  os << line << std::string(w + 70, '-') << "\n";
  for (const auto& r : results) {
    if (!r.error.empty()) {
      os << r.name << "  ERROR: " << r.error << "\n";
      continue;
    }
    std::snprintf(line, sizeof line, "%-*s %12.2f %12.2f %8lld %11.1f %10.1f %8s\n", int(w), r.name.c_str(),
                  r.avg_event_ns() / 1e3, r.best_event_ns / 1e3, (long long)r.iterations, r.gflops(), r.gbytes(),
                  r.kernel.empty() ? "-" : r.kernel.c_str());
    os << line;
  }
}

void print_csv(std::ostream& os, const std::vector<Result>& results) {
  os << "name,iterations,avg_event_time,best_event_time,avg_overall_time,best_overall_time,n_fl_ops,"
        "bytes_processed,gflops,gbytes_per_s,kernel,error_message\n";
  for (const auto& r : results) {
    os << '"' << r.name << "\"," << r.iterations << "," << r.avg_event_ns() << "," << r.best_event_ns << ","
       << r.avg_overall_ns() << "," << r.best_overall_ns << "," << r.n_fl_ops << "," << r.bytes_processed << ","
       << r.gflops() << "," << r.gbytes() << "," << r.kernel << ",\"" << r.error << "\"\n";
  }
}

void print_json(std::ostream& os, const std::vector<Result>& results, const Args& args, int compute_units) {
  os << "{\n  \"context\": {\"library\": \"libsbamd\", \"operator\": \"" << args.op << "\", \"dtype\": \"" << args.dtype
     << "\", \"device\": " << args.device << ", \"compute_units\": " << compute_units << ", \"kernel\": \""
     << args.kernel << "\", \"time_unit\": \"ns\"},\n  \"benchmarks\": [\n";
  for (size_t i = 0; i < results.size(); ++i) {
    const auto& r = results[i];
    os << "    {\"name\": \"" << r.name << "\", \"iterations\": " << r.iterations;
    if (!r.error.empty()) {
      os << ", \"error_occurred\": true, \"error_message\": \"" << r.error << "\"}";
    } else {
      os << ", \"avg_event_time\": " << r.avg_event_ns() << ", \"best_event_time\": " << r.best_event_ns
         << ", \"avg_overall_time\": " << r.avg_overall_ns() << ", \"best_overall_time\": " << r.best_overall_ns
         << ", \"total_event_time\": " << r.total_event_ns << ", \"total_overall_time\": " << r.total_overall_ns
         << ", \"n_fl_ops\": " << r.n_fl_ops << ", \"bytes_processed\": " << r.bytes_processed
         << ", \"gflops\": " << r.gflops() << ", \"gbytes_per_s\": " << r.gbytes() << ", \"kernel\": \"" << r.kernel
         << "\"}";
    }
    os << (i + 1 < results.size() ? ",\n" : "\n");
  }
  os << "  ]\n}\n";
}

void emit(std::ostream& os, const std::string& format, const std::vector<Result>& results, const Args& args, int cus) {
  if (format == "json") print_json(os, results, args, cus);
  else if (format == "csv") print_csv(os, results);
  else print_console(os, results);
}

}  // namespace

int main(int argc, char** argv) {
  try {
    Args args = parse_args(argc, argv);
    static const char* ops[] = {"gemm", "gemm_batched", "gemm_batched_strided", "symm", "trsm"};
    if (std::find(std::begin(ops), std::end(ops), args.op) == std::end(ops)) usage(2);
    const DType dt = dtype_of(args.dtype);
    const bool is_gemm = args.op.compare(0, 4, "gemm") == 0;
    if (!is_gemm && dt.fill_code > 1) throw std::runtime_error(args.op + " exists for float and double only");

    std::vector<Row> rows;
    if (args.csv_param.empty()) {
      std::fprintf(stderr, "WARNING: no CSV parameter file specified. Default ranges will be used.\n");
      rows = default_rows(args.op);
    } else {
      rows = rows_from_csv(args.op, args.csv_param);
    }
    std::vector<std::pair<std::string, Row>> selected;
    const std::regex filter(args.filter.empty() ? ".*" : args.filter);
    for (const Row& r : rows) {
      const std::string name = row_name(args.op, dt, r);
      if (std::regex_search(name, filter)) selected.emplace_back(name, r);
    }
    if (args.list_only) {
      for (const auto& s : selected) std::cout << s.first << "\n";
      return 0;
    }

    sbamd_handle_t h = nullptr;
    check(sbamd_create(&h, args.device, nullptr), nullptr, "sbamd_create");
    int cus = 0;
    sbamd_get_num_compute_units(h, &cus);
    check(sbamd_set_kernel(h, kernel_code(args.kernel), 0), h, "sbamd_set_kernel");

    std::vector<Result> results;
    for (const auto& s : selected) {
      Result r;
      try {
        if (is_gemm) r = run_gemm_row(h, args, dt, s.second);
        else if (dt.fill_code == 0) r = run_symm_trsm_row<float>(h, args, dt, s.second);
        else r = run_symm_trsm_row<double>(h, args, dt, s.second);
      } catch (const std::exception& e) {
        r.name = s.first;
        r.error = e.what();
      }
      results.push_back(r);
      if (args.format == "console") std::fprintf(stderr, "done %s\n", s.first.c_str());
    }
    sbamd_destroy(h);

    emit(std::cout, args.format, results, args, cus);
    if (!args.out_path.empty()) {
      std::ofstream out(args.out_path);
      if (!out) throw std::runtime_error("could not write " + args.out_path);
      emit(out, args.out_format, results, args, cus);
    }
    for (const auto& r : results)
      if (!r.error.empty()) return 1;
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "bench_blas3: %s\n", e.what());
    return 2;
  }
}
