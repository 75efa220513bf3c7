// host_utils.cpp -- see host_utils.hpp for the reference interfaces mirrored here.
#include "host_utils.hpp"
#include "options.hpp"

#include <getopt.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <thread>

namespace tkspmv {

// =====================================================================================================
// Options (same getopt string, long names, defaults and quirks as options.hpp:62-132)
// =====================================================================================================
const char *Options::default_matrix() {
    return "../../data/matrices_for_testing/matrices_small/matrix_1000_512_20_gamma.mtx";
}

Options::Options(int argc, char *argv[]) {
    static struct option long_options[] = {{"debug", no_argument, nullptr, 'd'},
                                           {"use_sample_matrix", no_argument, nullptr, 's'},
                                           {"no_reset", no_argument, nullptr, 'r'},
                                           {"matrix_path", required_argument, nullptr, 'm'},
                                           {"num_tests", required_argument, nullptr, 't'},
                                           {"xclbin", required_argument, nullptr, 'x'},
                                           {"ignore_matrix_values", no_argument, nullptr, 'v'},
                                           {"k", required_argument, nullptr, 'k'},
                                           {"block_size_1d", required_argument, nullptr, 'b'},
                                           {"block_size_2d", required_argument, nullptr, 'c'},
                                           {"num_blocks", required_argument, nullptr, 'g'},
                                           {"gpu_impl", required_argument, nullptr, 'i'},
                                           {"half_precision_gpu", no_argument, nullptr, 'a'},
                                           {nullptr, 0, nullptr, 0}};
    optind = 1;  // allow repeated parsing inside one process (tests)
    int opt, option_index = 0;
    while ((opt = getopt_long(argc, argv, "dm:st:x:vk:rb:c:g:i:a", long_options, &option_index)) != EOF) {
        switch (opt) {
            case 'd': debug = 1; break;
            case 'r': reset = true; break;  // sic: "--no_reset" keeps reset on, as the reference does
            case 'm': matrix_path = optarg; break;
            case 's': use_sample_matrix = true; break;
            case 't': num_tests = (unsigned)atoi(optarg); break;
            case 'x': xclbin_path = optarg; break;
            case 'v': ignore_matrix_values = true; break;
            case 'k': top_k_value = atoi(optarg); break;
            case 'b': block_size_1d = atoi(optarg); break;
            case 'c': block_size_2d = atoi(optarg); break;
            case 'g': num_blocks = atoi(optarg); break;
            case 'i': gpu_impl = atoi(optarg); break;
            case 'a': use_half_precision_gpu = true; break;
            default: break;
        }
    }
}

// =====================================================================================================
// MatrixMarket reader
// =====================================================================================================
namespace {

struct Cursor {
    const char *p, *end;
    bool eof() const { return p >= end; }
    void skip_ws() {
        while (p < end && (unsigned char)*p <= ' ') ++p;
    }
    // Returns the current line [b, e) without the newline and advances past it. false at EOF.
    bool line(const char *&b, const char *&e) {
        if (p >= end) return false;
        b = p;
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        e = nl ? nl : end;
        p = nl ? nl + 1 : end;
        return true;
    }
};

inline bool parse_u32(Cursor &c, uint32_t &out) {
    c.skip_ws();
    if (c.eof()) return false;
    const char *s = c.p;
    if (*s == '+') ++s;
    uint64_t v = 0;
    const char *d0 = s;
    while (s < c.end && *s >= '0' && *s <= '9') {
        v = v * 10 + (uint64_t)(*s - '0');
        ++s;
    }
    if (s == d0) return false;
    c.p = s;
    out = (uint32_t)v;
    return true;
}

const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                           1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// Correctly rounded decimal -> double for the common short form (Clinger's fast path: <= 15 significant
// digits and |exp10| <= 22 => one exact multiplication or division); strtod otherwise. Matches "%lf".
inline bool parse_double(Cursor &c, double &out) {
    c.skip_ws();
    if (c.eof()) return false;
    const char *s = c.p;
    bool neg = false;
    if (*s == '-' || *s == '+') {
        neg = (*s == '-');
        ++s;
    }
    uint64_t mant = 0;
    int digits = 0, exp10 = 0;
    bool any = false, fast = true;
    while (s < c.end && *s >= '0' && *s <= '9') {
        any = true;
        if (mant || *s != '0') {
            if (digits < 18) {
                mant = mant * 10 + (uint64_t)(*s - '0');
                ++digits;
            } else {
                fast = false;
                ++exp10;
            }
        }
        ++s;
    }
    if (s < c.end && *s == '.') {
        ++s;
        while (s < c.end && *s >= '0' && *s <= '9') {
            any = true;
            if (mant || *s != '0') {
                if (digits < 18) {
                    mant = mant * 10 + (uint64_t)(*s - '0');
                    ++digits;
                    --exp10;
                } else {
                    fast = false;
                }
            } else {
                --exp10;
            }
            ++s;
        }
    }
    if (!any) {  // inf / nan / garbage: defer to strtod
        char *endp = nullptr;
        double v = strtod(c.p, &endp);
        if (endp == c.p) return false;
        c.p = endp;
        out = v;
        return true;
    }
    if (s < c.end && (*s == 'e' || *s == 'E')) {
        const char *t = s + 1;
        bool eneg = false;
        if (t < c.end && (*t == '-' || *t == '+')) {
            eneg = (*t == '-');
            ++t;
        }
        if (t < c.end && *t >= '0' && *t <= '9') {
            int ev = 0;
            while (t < c.end && *t >= '0' && *t <= '9') {
                if (ev < 100000) ev = ev * 10 + (*t - '0');
                ++t;
            }
            exp10 += eneg ? -ev : ev;
            s = t;
        }
    }
    if (fast && digits <= 15 && exp10 >= -22 && exp10 <= 22) {
        double v = (double)mant;
        v = exp10 < 0 ? v / kPow10[-exp10] : v * kPow10[exp10];
        out = neg ? -v : v;
        c.p = s;
        return true;
    }
    // Slow path: let libc round it. The token is bounded by whitespace, and the buffer is NUL-terminated.
    char *endp = nullptr;
    double v = strtod(c.p, &endp);
    if (endp == c.p) return false;
    c.p = endp;
    out = v;
    return true;
}

std::string lower(std::string s) {
    for (char &ch : s) ch = (char)tolower((unsigned char)ch);
    return s;
}

}  // namespace

void sort_coo(CooMatrix &m) {
    size_t n = m.row.size();
    std::vector<uint64_t> perm(n);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](uint64_t a, uint64_t b) {
        if (m.row[a] != m.row[b]) return m.row[a] < m.row[b];
        return m.col[a] < m.col[b];
    });
    std::vector<uint32_t> r(n), c(n);
    std::vector<float> v(n);
    for (size_t i = 0; i < n; ++i) {
        r[i] = m.row[perm[i]];
        c[i] = m.col[perm[i]];
        v[i] = m.val[perm[i]];
    }
    m.row.swap(r);
    m.col.swap(c);
    m.val.swap(v);
}

IoError read_mtx(const std::string &path, int index_base, bool read_values, bool sort, CooMatrix &out) {
    out = CooMatrix();
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return {5, "File " + path + " not found"};
    std::vector<char> buf;
    {
        fseek(f, 0, SEEK_END);
        long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        if (sz < 0) sz = 0;
        buf.resize((size_t)sz + 1);
        size_t got = sz ? fread(buf.data(), 1, (size_t)sz, f) : 0;
        buf[got] = '\0';
        buf.resize(got + 1);
        fclose(f);
    }
    Cursor c{buf.data(), buf.data() + buf.size() - 1};

    // Banner: five tokens, "%%MatrixMarket matrix coordinate <type> <storage>".
    const char *lb, *le;
    const IoError bad_banner{5, "Could not process Matrix Market banner."};
    if (!c.line(lb, le)) return bad_banner;
    {
        std::string line(lb, le);
        char t[5][65];
        if (sscanf(line.c_str(), "%64s %64s %64s %64s %64s", t[0], t[1], t[2], t[3], t[4]) != 5) return bad_banner;
        if (strncmp(t[0], "%%MatrixMarket", 14) != 0) return bad_banner;
        std::string mtx = lower(t[1]), crd = lower(t[2]), dt = lower(t[3]), st = lower(t[4]);
        if (mtx != "matrix") return bad_banner;
        if (crd != "coordinate") {
            if (crd == "array") return {6, "dense 'array' MatrixMarket files are not supported"};
            return bad_banner;
        }
        bool pattern = false;
        if (dt == "real" || dt == "integer")
            pattern = false;
        else if (dt == "pattern")
            pattern = true;
        else if (dt == "complex")
            return {6, "complex MatrixMarket files are not supported"};
        else
            return bad_banner;
        if (st == "general")
            out.symmetric = false;
        else if (st == "symmetric" || st == "hermitian" || st == "skew-symmetric")
            out.symmetric = (st == "symmetric");
        else
            return bad_banner;
        if (pattern) read_values = false;
        // Reference quirk kept: with read_values == false and a real/integer file the value token is NOT
        // consumed by the reference (utils.hpp:383-387) and parsing derails; here the token is consumed
        // and the value forced to 1, which is what `-v` means.
        // Size line: skip comment lines, then "%u %u %u"; keep scanning tokens if that line is blank.
        uint32_t M = 0, N = 0, NZ = 0;
        bool have_size = false;
        while (c.line(lb, le)) {
            if (lb < le && *lb == '%') continue;
            std::string sl(lb, le);
            if (sscanf(sl.c_str(), "%u %u %u", &M, &N, &NZ) == 3) {
                have_size = true;
            } else {
                // blank or partial line: the reference falls back to fscanf over the remaining stream
                Cursor t2{lb, c.end};
                if (parse_u32(t2, M) && parse_u32(t2, N) && parse_u32(t2, NZ)) {
                    have_size = true;
                    c.p = t2.p;
                }
            }
            break;
        }
        if (!have_size) return {5, "Could not read the MatrixMarket size line (premature EOF)."};
        out.rows = M;
        out.cols = N;
        out.row.resize(NZ);
        out.col.resize(NZ);
        out.val.resize(NZ);
        const bool has_value_token = !pattern;
        uint32_t min_idx = 0xFFFFFFFFu, max_r = 0, max_c = 0;
        for (uint32_t i = 0; i < NZ; ++i) {
            uint32_t r, cc;
            double v = 1.0;
            if (!parse_u32(c, r)) return {5, "Error: Not enough rows in mtx file!"};
            if (!parse_u32(c, cc)) return {5, "Error: Not enough rows in mtx file!"};
            if (has_value_token) {
                if (!parse_double(c, v)) return {5, "Error: malformed value in mtx file"};
                if (!read_values) v = 1.0;
            }
            out.row[i] = r;
            out.col[i] = cc;
            out.val[i] = (float)v;
            min_idx = std::min(min_idx, std::min(r, cc));
            max_r = std::max(max_r, r);
            max_c = std::max(max_c, cc);
        }
        int base = index_base;
        if (base < 0) {
            if (NZ == 0)
                base = 0;
            else
                base = (min_idx == 0) ? 0 : 1;
        }
        out.index_base = base;
        if (base == 1) {
            if (NZ && min_idx == 0) return {1, "index_base=1 requested but the file contains index 0"};
            for (uint32_t i = 0; i < NZ; ++i) {
                --out.row[i];
                --out.col[i];
            }
        }
    }
    if (out.symmetric) {  // undirect(): mirror off-diagonal entries (utils.hpp:406-419), then order them
        size_t n = out.row.size();
        for (size_t i = 0; i < n; ++i) {
            if (out.col[i] != out.row[i]) {
                out.row.push_back(out.col[i]);
                out.col.push_back(out.row[i]);
                out.val.push_back(out.val[i]);
            }
        }
        sort = true;  // the reference leaves them appended (and then violates its own row-major assumption)
    }
    if (sort) sort_coo(out);
    uint32_t mr = 0;
    for (uint32_t r : out.row) mr = std::max(mr, r);
    out.num_rows_coo = out.row.empty() ? 0 : mr + 1;
    return {};
}

IoError write_mtx(const std::string &path, uint32_t rows, uint32_t cols, uint64_t nnz, const uint32_t *row,
                  const uint32_t *col, const float *val, int index_base, int precision) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return {5, "cannot open " + path + " for writing"};
    if (precision <= 0) precision = 10;  // DEFAULT_PRECISION, create_matrices.py:28
    // Header exactly as create_matrices.py:33
    fprintf(f, "%%%%MatrixMarket matrix coordinate real general\n%%\n%u %u %llu\n", rows, cols,
            (unsigned long long)nnz);
    std::vector<char> buf(1 << 20);
    setvbuf(f, buf.data(), _IOFBF, buf.size());
    for (uint64_t i = 0; i < nnz; ++i)
        fprintf(f, "%u %u %.*g\n", row[i] + (uint32_t)index_base, col[i] + (uint32_t)index_base, precision,
                (double)(val ? val[i] : 1.0f));
    fclose(f);
    return {};
}

// =====================================================================================================
// Query vector (create_sample_vector, utils.hpp:234-267): mt19937 + uniform_real_distribution<double>
// =====================================================================================================
void sample_vector(float *vec, int size, bool random, bool sum_to_one, bool norm_one, int seed) {
    if (random) {
        std::random_device rd;
        std::mt19937 engine(seed == 0 ? rd() : (unsigned)seed);
        std::uniform_real_distribution<double> dist(0, 1);
        for (int i = 0; i < size; ++i) vec[i] = (float)dist(engine);
    } else {
        for (int i = 0; i < size; ++i) vec[i] = 1.0f;
    }
    if (sum_to_one) {
        float sum = 0;
        for (int i = 0; i < size; ++i) sum += vec[i];
        for (int i = 0; i < size; ++i) vec[i] = vec[i] / sum;
    } else if (norm_one) {
        double sum = 0;
        for (int i = 0; i < size; ++i) sum += vec[i] * vec[i];  // float product, double accumulation
        const double root = std::sqrt(sum);
        for (int i = 0; i < size; ++i) vec[i] = (float)(vec[i] / root);
    }
}

// =====================================================================================================
// Synthetic matrices. Distributions restate create_matrices.py:83-104; the PRNG is ours (xoshiro256**
// keyed per row so generation is reproducible and order-independent). Seeds are ours: the reference is
// unseeded.
// =====================================================================================================
namespace {
struct Rng {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t &x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    Rng(uint64_t seed, uint64_t stream) {
        uint64_t x = seed * 0xD1342543DE82EF95ull + stream * 0x2545F4914F6CDD1Dull + 0x1234567ull;
        for (auto &v : s) v = splitmix(x);
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }  // [0,1)
    double uniform_open() { return ((double)(next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }  // (0,1)
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
};
}  // namespace

namespace {
uint32_t row_degree(uint32_t r, uint32_t avg_nnz, int dist, uint64_t seed) {
    Rng g(seed, (uint64_t)r * 2);
    if (dist == DIST_UNIFORM) {
        uint32_t lo = avg_nnz / 2, hi = (uint32_t)(avg_nnz * 1.5);  // randint(min, max + 1) => inclusive
        return lo + g.below(hi - lo + 1);
    }
    // Gamma(shape=3, scale=avg/3) as an Erlang-3 variate; np.maximum(int(.), 1)
    double u = g.uniform_open() * g.uniform_open() * g.uniform_open();
    double x = -std::log(u) * ((double)avg_nnz / 3.0);
    uint32_t d = (uint32_t)x;
    return d < 1 ? 1u : d;
}
// Rows [r0, r1) of the matrix into out_* at the positions their degrees give; row ids are written as r - row_base.
void fill_rows(uint32_t r0, uint32_t r1, uint32_t cols, uint64_t seed, const uint32_t *deg, const uint64_t *pos_of,
               uint32_t row_base, uint32_t *orow, uint32_t *ocol, float *oval) {
    std::vector<uint32_t> cbuf;
    std::vector<double> vbuf;
    for (uint32_t r = r0; r < r1; ++r) {
        Rng g(seed, (uint64_t)r * 2 + 1);
        const uint32_t d = deg[r - row_base];
        uint64_t pos = pos_of[r - row_base];
        cbuf.resize(d);
        vbuf.resize(d);
        for (uint32_t j = 0; j < d; ++j) cbuf[j] = g.below(cols);  // with replacement => duplicates happen
        std::sort(cbuf.begin(), cbuf.end());
        double n2 = 0;
        for (uint32_t j = 0; j < d; ++j) {
            vbuf[j] = g.uniform();
            n2 += vbuf[j] * vbuf[j];
        }
        const double inv = n2 > 0 ? 1.0 / std::sqrt(n2) : 0.0;
        for (uint32_t j = 0; j < d; ++j, ++pos) {
            orow[pos] = r - row_base;
            ocol[pos] = cbuf[j];
            oval[pos] = (float)(vbuf[j] * inv);
        }
    }
}
unsigned host_threads() {
    unsigned n = std::thread::hardware_concurrency();
    if (const char *e = opt("HOST_THREADS")) n = (unsigned)atoi(e);
    return n < 1 ? 1u : (n > 64 ? 64u : n);
}
template <class F>
void parallel_rows(uint32_t r0, uint32_t r1, F f) {
    const unsigned nt = (r1 - r0) < 65536u ? 1u : host_threads();
    if (nt == 1) {
        f(r0, r1);
        return;
    }
    std::vector<std::thread> th;
    const uint64_t n = r1 - r0;
    for (unsigned t = 0; t < nt; ++t) {
        const uint32_t a = r0 + (uint32_t)(n * t / nt), b = r0 + (uint32_t)(n * (t + 1) / nt);
        if (a < b) th.emplace_back([=]() { f(a, b); });
    }
    for (auto &t : th) t.join();
}
}  // namespace

// Every row has its own PRNG streams (degree: stream 2r, content: stream 2r + 1), so any row range of the matrix can be
// generated on its own, in any order and on any number of threads, with identical results.
void generate_degrees(uint32_t row_begin, uint32_t row_end, uint32_t avg_nnz, int dist, uint64_t seed, uint32_t *deg) {
    parallel_rows(row_begin, row_end, [=](uint32_t a, uint32_t b) {
        for (uint32_t r = a; r < b; ++r) deg[r - row_begin] = row_degree(r, avg_nnz, dist, seed);
    });
}

void generate_matrix_rows(uint32_t row_begin, uint32_t row_end, uint32_t cols, uint32_t avg_nnz, int dist, uint64_t seed,
                          CooMatrix &out) {
    out = CooMatrix();
    const uint32_t n = row_end > row_begin ? row_end - row_begin : 0u;
    out.rows = n;
    out.cols = cols;
    out.index_base = 0;
    std::vector<uint32_t> deg(n);
    generate_degrees(row_begin, row_begin + n, avg_nnz, dist, seed, deg.data());
    std::vector<uint64_t> pos(n + 1, 0);
    for (uint32_t i = 0; i < n; ++i) pos[i + 1] = pos[i] + deg[i];
    const uint64_t total = pos[n];
    out.row.resize(total);
    out.col.resize(total);
    out.val.resize(total);
    uint32_t *orow = out.row.data(), *ocol = out.col.data();
    float *oval = out.val.data();
    const uint32_t *dp = deg.data();
    const uint64_t *pp = pos.data();
    parallel_rows(row_begin, row_begin + n, [=](uint32_t a, uint32_t b) {
        fill_rows(a, b, cols, seed, dp, pp, row_begin, orow, ocol, oval);
    });
    out.num_rows_coo = n;
}

void generate_matrix(uint32_t rows, uint32_t cols, uint32_t avg_nnz, int dist, uint64_t seed, CooMatrix &out) {
    generate_matrix_rows(0, rows, cols, avg_nnz, dist, seed, out);
}

// =====================================================================================================
// Evaluation helpers
// =====================================================================================================
void sort_tuples(size_t n, uint32_t *idx, float *val) {
    std::vector<std::pair<uint32_t, float>> t(n);
    for (size_t i = 0; i < n; ++i) t[i] = {idx[i], val[i]};
    std::sort(t.begin(), t.end(), [](const std::pair<uint32_t, float> &l, const std::pair<uint32_t, float> &r) {
        if (l.second != r.second) return l.second > r.second;
        return l.first > r.first;
    });
    for (size_t i = 0; i < n; ++i) {
        idx[i] = t[i].first;
        val[i] = t[i].second;
    }
}

int check_array_equality(const float *x, const float *y, int n, float tol) {
    int errors = 0;
    for (int i = 0; i < n; ++i) {
        float diff = (x[i] > y[i]) ? (x[i] - y[i]) : (y[i] - x[i]);
        if (diff > tol) ++errors;
    }
    return errors;
}

int check_array_equality(const uint32_t *x, const uint32_t *y, int n) {
    int errors = 0;
    for (int i = 0; i < n; ++i) errors += (x[i] != y[i]);
    return errors;
}

float mean(const std::vector<float> &x, int skip) {
    int fixed = (int)x.size() - skip;
    if (fixed <= 0) return 0.0f;
    float sum = 0;
    for (size_t i = (size_t)skip; i < x.size(); ++i) sum += x[i];
    return sum / (float)fixed;
}

float st_dev(const std::vector<float> &x, int skip) {
    int fixed = (int)x.size() - skip;
    if (fixed <= 0) return 0.0f;
    float m = 0, m2 = 0;
    for (size_t i = (size_t)skip; i < x.size(); ++i) {
        m += x[i];
        m2 += x[i] * x[i];
    }
    float diff = m2 - m * m / (float)fixed;
    if (diff < 0) diff = 0;
    return std::sqrt(diff / (float)fixed);
}

// The same gold with its row sums formed by several threads (round 5: the reference's grid reaches 600M non-zeros, 6 s per query on
// one core). The result is the single-threaded one's, list for list: the entry range is cut at row boundaries, every thread forms the
// sums of its runs of equal row ids in entry order -- the same fp32 additions in the same order --, and ONE thread then offers the
// (row, sum) pairs to the list in the original order with the original rule.
static void gold_topk_threaded(const uint32_t *row, const uint32_t *col, const float *val, uint64_t nnz, const float *vec, int k,
                               uint32_t *res_idx, float *res_val, unsigned nt) {
    std::vector<uint64_t> cut(nt + 1);
    for (unsigned t = 0; t <= nt; ++t) {
        uint64_t c = nnz * t / nt;
        while (c > 0 && c < nnz && row[c] == row[c - 1]) ++c;  // (forward to the next change of row)
        cut[t] = c;
    }
    std::vector<std::vector<uint32_t>> rr(nt);
    std::vector<std::vector<float>> ss(nt);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&, t]() {
            const uint64_t a = cut[t], b = cut[t + 1];
            if (a >= b) return;
            uint32_t cur = row[a];
            float acc = 0.0f;
            for (uint64_t i = a; i < b; ++i) {
                const float contrib = val[i] * vec[col[i]];
                if (row[i] == cur) {
                    acc += contrib;
                } else {
                    rr[t].push_back(cur);
                    ss[t].push_back(acc);
                    cur = row[i];
                    acc = contrib;
                }
            }
            rr[t].push_back(cur);
            ss[t].push_back(acc);
        });
    for (auto &x : th) x.join();
    uint32_t worst_pos = 0;
    float worst_val = 0.0f;
    size_t left = 0;
    for (unsigned t = 0; t < nt; ++t) left += rr[t].size();
    for (unsigned t = 0; t < nt; ++t)
        for (size_t j = 0; j < rr[t].size(); ++j) {
            const float score = ss[t][j];
            --left;
            if (score >= worst_val) {
                res_idx[worst_pos] = rr[t][j];
                res_val[worst_pos] = score;
                if (left != 0) {  // (the reference does not re-scan after the last row)
                    uint32_t wp = 0;
                    float wv = res_val[0];
                    for (int q = 0; q < k; ++q)
                        if (res_val[q] < wv) {
                            wv = res_val[q];
                            wp = (uint32_t)q;
                        }
                    worst_pos = wp;
                    worst_val = wv;
                }
            }
        }
}

void gold_topk(const uint32_t *row, const uint32_t *col, const float *val, uint64_t nnz, const float *vec, int k,
               uint32_t *res_idx, float *res_val) {
    for (int i = 0; i < k; ++i) {
        res_idx[i] = 0;
        res_val[i] = 0.0f;
    }
    if (nnz == 0 || k <= 0) return;
    if (nnz >= (1ull << 25)) {  // (32M non-zeros and up: worth the threads)
        const unsigned nt = std::max(1u, std::min(host_threads(), 64u));
        if (nt > 1) {
            gold_topk_threaded(row, col, val, nnz, vec, k, res_idx, res_val, nt);
            return;
        }
    }
    uint32_t worst_pos = 0;
    float worst_val = 0.0f;
    auto offer = [&](uint32_t r, float score, bool rescan) {
        if (score >= worst_val) {
            res_idx[worst_pos] = r;
            res_val[worst_pos] = score;
            if (rescan) {  // first strict minimum wins, as a left-to-right scan with '<'
                uint32_t wp = 0;
                float wv = res_val[0];
                for (int j = 0; j < k; ++j)
                    if (res_val[j] < wv) {
                        wv = res_val[j];
                        wp = (uint32_t)j;
                    }
                worst_pos = wp;
                worst_val = wv;
            }
        }
    };
    uint32_t cur = row[0];
    float acc = 0.0f;
    for (uint64_t i = 0; i < nnz; ++i) {
        float contrib = val[i] * vec[col[i]];
        if (row[i] == cur) {
            acc += contrib;
        } else {
            offer(cur, acc, true);
            cur = row[i];
            acc = contrib;
        }
    }
    offer(cur, acc, false);  // the reference does not re-scan after the last row
}

}  // namespace tkspmv
