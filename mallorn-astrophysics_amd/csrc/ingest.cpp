// ingest.cpp -- liblcfe_ingest.so: light-curve CSV files -> CSR arrays (see include/lcfe_ingest.h).
//
// Host-only C++17 (no HIP).  The files are memory-mapped, cut into line-aligned chunks, the chunks
// parsed by a pool of threads into flat columns, and one sequential pass numbers the objects by
// first appearance and counts their rows; lcfe_csv_fill() scatters the rows into the caller's CSR
// arrays (file order inside each object).  Number conversion restates pandas' default C-parser
// converter (precise_xstrtod, pandas/_libs/src/parser/tokenizer.c) so that the arrays are bit-identical
// to ``pd.read_csv`` of the reference's loader (src/utils/data_loader.py:36-62).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/lcfe_ingest.h"

namespace {

thread_local std::string g_err;

// ---- pandas' precise_xstrtod ------------------------------------------------------------------
struct Pow10 {
    double e[309];
    Pow10() {
        for (int k = 0; k <= 308; ++k) {
            char buf[16];
            snprintf(buf, sizeof buf, "1e%d", k);
            e[k] = strtod(buf, nullptr);          // the correctly rounded literals pandas compiles in
        }
    }
};
const Pow10 kPow10;

inline bool is_digit(char c) { return c >= '0' && c <= '9'; }
inline bool is_space(char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

// Parses [p, end) the way precise_xstrtod does; returns the position after the number
// (nullptr: no digits).  At most 17 significant digits enter `number`; further integer digits
// only raise the exponent, further decimals are dropped.
const char* precise_number(const char* p, const char* end, double* out, bool* overflow) {
    constexpr int max_digits = 17;
    bool negative = false;
    if (p < end && (*p == '-' || *p == '+')) { negative = (*p == '-'); ++p; }
    double number = 0.0;
    int exponent = 0, num_digits = 0, num_decimals = 0;
    while (p < end && is_digit(*p)) {
        if (num_digits < max_digits) { number = number * 10.0 + (*p - '0'); ++num_digits; }
        else ++exponent;
        ++p;
    }
    if (p < end && *p == '.') {
        ++p;
        while (num_digits < max_digits && p < end && is_digit(*p)) {
            number = number * 10.0 + (*p - '0');
            ++p; ++num_digits; ++num_decimals;
        }
        if (num_digits >= max_digits)
            while (p < end && is_digit(*p)) ++p;
        exponent -= num_decimals;
    }
    if (num_digits == 0) return nullptr;
    if (negative) number = -number;
    if (p < end && (*p == 'e' || *p == 'E')) {
        const char* q = p + 1;
        bool eneg = false;
        if (q < end && (*q == '-' || *q == '+')) { eneg = (*q == '-'); ++q; }
        int n = 0, nd = 0;
        while (nd < max_digits && q < end && is_digit(*q)) { n = n * 10 + (*q - '0'); ++nd; ++q; }
        if (nd > 0) { exponent += eneg ? -n : n; p = q; }     // no digits after 'e': the 'e' is not consumed
    }
    *overflow = false;
    if (exponent > 308) { *overflow = true; number = HUGE_VAL; }
    else if (exponent > 0) number *= kPow10.e[exponent];
    else if (exponent < -308) {
        if (exponent < -616) number = 0.0;
        else { number /= kPow10.e[-308 - exponent]; number /= kPow10.e[308]; }
    } else number /= kPow10.e[-exponent];
    if (number == HUGE_VAL || number == -HUGE_VAL) *overflow = true;
    *out = number;
    return p;
}

bool ieq(const char* s, size_t n, const char* lit) {
    if (strlen(lit) != n) return false;
    for (size_t k = 0; k < n; ++k) {
        char c = s[k];
        if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
        if (c != lit[k]) return false;
    }
    return true;
}

// pandas' default NA strings (pandas/_libs/parsers.pyx STR_NA_VALUES)
bool is_na(const char* s, size_t n) {
    static const char* const kNa[] = {"", "#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND",
                                      "1.#QNAN", "<NA>", "N/A", "NA", "NULL", "NaN", "None", "n/a", "nan", "null"};
    for (const char* lit : kNa)
        if (strlen(lit) == n && memcmp(lit, s, n) == 0) return true;
    return false;
}

// One CSV field -> double.  0 = ok, 1 = not a number.
int field_to_double(const char* s, size_t n, double* out) {
    if (is_na(s, n)) { *out = std::nan(""); return 0; }
    const char* p = s;
    const char* end = s + n;
    while (p < end && is_space(*p)) ++p;
    bool overflow = false;
    const char* q = precise_number(p, end, out, &overflow);
    if (q) {
        while (q < end && is_space(*q)) ++q;
        if (q == end && !overflow) return 0;
    }
    // pandas retries the word as an infinity spelling
    const size_t m = (size_t)(end - p);
    if (ieq(p, m, "inf") || ieq(p, m, "infinity") || ieq(p, m, "+inf") || ieq(p, m, "+infinity")) { *out = HUGE_VAL; return 0; }
    if (ieq(p, m, "-inf") || ieq(p, m, "-infinity")) { *out = -HUGE_VAL; return 0; }
    return 1;
}

// ---- files and chunks ---------------------------------------------------------------------------
struct Mapped {
    const char* data = nullptr;
    size_t size = 0;
    int fd = -1;
};

struct Columns { int id = -1, t = -1, f = -1, e = -1, b = -1, n = 0; };

struct Chunk {
    const char* begin;
    const char* end;
    Columns cols;
    int file;
    std::vector<double> t, f, e;
    std::vector<uint8_t> band;
    std::vector<const char*> id_ptr;
    std::vector<uint32_t> id_len;
    std::string err;
};

// field of the line [p, line_end) that starts at p (p <= line_end): sets [fs, fe) to its unquoted
// content and returns the position of its delimiter (a ',' or line_end); nullptr on a malformed quote
const char* field_at(const char* p, const char* line_end, const char** fs, const char** fe) {
    if (p < line_end && *p == '"') {
        const char* q = (const char*)memchr(p + 1, '"', (size_t)(line_end - p - 1));
        if (!q || (q + 1 < line_end && q[1] != ',')) return nullptr;     // escaped quotes are not supported
        *fs = p + 1;
        *fe = q;
        return q + 1;
    }
    const char* q = (p < line_end) ? (const char*)memchr(p, ',', (size_t)(line_end - p)) : nullptr;
    *fs = p;
    *fe = q ? q : line_end;
    return *fe;
}

uint8_t band_code(const char* s, size_t n) {
    if (n != 1) return 255;
    switch (*s) {
        case 'u': return 0; case 'g': return 1; case 'r': return 2;
        case 'i': return 3; case 'z': return 4; case 'y': return 5;
        default: return 255;
    }
}

void parse_chunk(Chunk& c) {
    const size_t guess = (size_t)(c.end - c.begin) / 48 + 16;
    c.t.reserve(guess); c.f.reserve(guess); c.e.reserve(guess); c.band.reserve(guess);
    c.id_ptr.reserve(guess); c.id_len.reserve(guess);
    const char* p = c.begin;
    while (p < c.end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(c.end - p));
        const char* line_end = nl ? nl : c.end;
        const char* next = nl ? nl + 1 : c.end;
        if (line_end > p && line_end[-1] == '\r') --line_end;
        if (line_end == p) { p = next; continue; }             // blank line (pandas skips them)
        const char* q = p;
        double tv = 0, fv = 0, ev = 0;
        const char* ids = nullptr; size_t idn = 0;
        uint8_t bc = 255;
        int have = 0;
        for (int col = 0;; ++col) {
            const char *fs, *fe;
            const char* d = field_at(q, line_end, &fs, &fe);
            if (!d) { c.err = "malformed quoted field"; return; }
            const size_t n = (size_t)(fe - fs);
            int bad = 0;
            if (col == c.cols.t) { bad = field_to_double(fs, n, &tv); ++have; }
            else if (col == c.cols.f) { bad = field_to_double(fs, n, &fv); ++have; }
            else if (col == c.cols.e) { bad = field_to_double(fs, n, &ev); ++have; }
            else if (col == c.cols.b) { bc = band_code(fs, n); ++have; }
            else if (col == c.cols.id) { ids = fs; idn = n; ++have; }
            if (bad) { c.err = "not a number: '" + std::string(fs, n > 40 ? 40 : n) + "'"; return; }
            if (d >= line_end) break;
            q = d + 1;
        }
        if (have != 5) { c.err = "row with missing columns"; return; }
        c.t.push_back(tv); c.f.push_back(fv); c.e.push_back(ev); c.band.push_back(bc);
        c.id_ptr.push_back(ids); c.id_len.push_back((uint32_t)idn);
        p = next;
    }
}

}  // namespace

struct lcfe_csv {
    std::vector<Mapped> files;
    std::vector<Chunk> chunks;
    std::vector<int32_t> obj_of_row;           // global row order = chunk order
    std::vector<int64_t> counts;
    std::vector<std::string_view> ids;
    int64_t n_rows = 0;
    int64_t id_bytes = 0;
    ~lcfe_csv() {
        for (auto& m : files) {
            if (m.data) munmap((void*)m.data, m.size);
            if (m.fd >= 0) close(m.fd);
        }
    }
};

extern "C" {

const char* lcfe_ingest_last_error(void) { return g_err.c_str(); }

int lcfe_csv_parse_double(const char* s, size_t len, double* out) { return field_to_double(s, len, out); }

lcfe_csv* lcfe_csv_open(const char* const* paths, int n_paths, int n_threads) {
    g_err.clear();
    if (!paths || n_paths <= 0) { g_err = "lcfe_csv_open: no paths"; return nullptr; }
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads < 1) n_threads = 1;
    auto* h = new lcfe_csv;
    auto fail = [&](const std::string& m) -> lcfe_csv* { g_err = m; delete h; return nullptr; };
    h->files.resize(n_paths);
    constexpr size_t kChunk = 8u << 20;
    for (int k = 0; k < n_paths; ++k) {
        Mapped& m = h->files[k];
        m.fd = open(paths[k], O_RDONLY);
        if (m.fd < 0) return fail(std::string("lcfe_csv_open: cannot open ") + paths[k]);
        struct stat st;
        if (fstat(m.fd, &st) != 0 || st.st_size <= 0) return fail(std::string("lcfe_csv_open: empty or unreadable ") + paths[k]);
        m.size = (size_t)st.st_size;
        void* a = mmap(nullptr, m.size, PROT_READ, MAP_PRIVATE, m.fd, 0);
        if (a == MAP_FAILED) { m.data = nullptr; return fail(std::string("lcfe_csv_open: mmap failed for ") + paths[k]); }
        m.data = (const char*)a;
        // header
        const char* end = m.data + m.size;
        const char* nl = (const char*)memchr(m.data, '\n', m.size);
        const char* he = nl ? nl : end;
        const char* body = nl ? nl + 1 : end;
        if (he > m.data && he[-1] == '\r') --he;
        Columns cols;
        const char* p = m.data;
        if (m.size >= 3 && (unsigned char)p[0] == 0xEF && (unsigned char)p[1] == 0xBB && (unsigned char)p[2] == 0xBF) p += 3;   // BOM
        for (int col = 0;; ++col) {
            const char *fs, *fe;
            const char* d = field_at(p, he, &fs, &fe);
            if (!d) return fail(std::string("lcfe_csv_open: malformed header in ") + paths[k]);
            const std::string_view name(fs, (size_t)(fe - fs));
            if (name == "object_id") cols.id = col;
            else if (name == "Time (MJD)") cols.t = col;
            else if (name == "Flux") cols.f = col;
            else if (name == "Flux_err") cols.e = col;
            else if (name == "Filter") cols.b = col;
            cols.n = col + 1;
            if (d >= he) break;
            p = d + 1;
        }
        if (cols.id < 0 || cols.t < 0 || cols.f < 0 || cols.e < 0 || cols.b < 0)
            return fail(std::string("lcfe_csv_open: a column of object_id / Time (MJD) / Flux / Flux_err / Filter is missing in ") + paths[k]);
        // line-aligned chunks
        const char* c0 = body;
        while (c0 < end) {
            const char* c1 = (size_t)(end - c0) > kChunk ? c0 + kChunk : end;
            if (c1 < end) {
                const char* q = (const char*)memchr(c1, '\n', (size_t)(end - c1));
                c1 = q ? q + 1 : end;
            }
            Chunk c;
            c.begin = c0; c.end = c1; c.cols = cols; c.file = k;
            h->chunks.push_back(std::move(c));
            c0 = c1;
        }
    }
    // parse the chunks
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= h->chunks.size()) return;
            parse_chunk(h->chunks[k]);
        }
    };
    {
        std::vector<std::thread> pool;
        const int nt = (int)std::min<size_t>((size_t)n_threads, h->chunks.size() ? h->chunks.size() : 1);
        for (int k = 1; k < nt; ++k) pool.emplace_back(worker);
        worker();
        for (auto& th : pool) th.join();
    }
    for (auto& c : h->chunks)
        if (!c.err.empty()) return fail(std::string("lcfe_csv_open: ") + c.err + " in " + paths[c.file]);
    // number the objects by first appearance (rows of one object are usually contiguous: one
    // comparison with the previous row's id, a hash lookup otherwise)
    int64_t total = 0;
    for (auto& c : h->chunks) total += (int64_t)c.t.size();
    if (total > 0x7fffffff) return fail("lcfe_csv_open: more than 2^31 rows");
    h->n_rows = total;
    h->obj_of_row.resize((size_t)total);
    std::unordered_map<std::string_view, int32_t> index;
    index.reserve(1 << 16);
    std::string_view prev;
    int32_t prev_idx = -1;
    size_t row = 0;
    for (auto& c : h->chunks) {
        const size_t n = c.t.size();
        for (size_t k = 0; k < n; ++k, ++row) {
            const std::string_view id(c.id_ptr[k], c.id_len[k]);
            if (prev_idx < 0 || id != prev) {
                auto it = index.find(id);
                if (it == index.end()) {
                    it = index.emplace(id, (int32_t)h->ids.size()).first;
                    h->ids.push_back(id);
                    h->counts.push_back(0);
                    h->id_bytes += (int64_t)id.size();
                }
                prev = id;
                prev_idx = it->second;
            }
            h->obj_of_row[row] = prev_idx;
            ++h->counts[(size_t)prev_idx];
        }
    }
    return h;
}

void lcfe_csv_close(lcfe_csv* h) { delete h; }

int64_t lcfe_csv_n_objects(const lcfe_csv* h) { return h ? (int64_t)h->ids.size() : 0; }
int64_t lcfe_csv_n_rows(const lcfe_csv* h) { return h ? h->n_rows : 0; }
int64_t lcfe_csv_id_bytes(const lcfe_csv* h) { return h ? h->id_bytes : 0; }

int lcfe_csv_fill(const lcfe_csv* h, int64_t* offsets, double* t, double* flux, double* err, uint8_t* band,
                  int64_t* id_offsets, char* id_bytes) {
    g_err.clear();
    if (!h || !offsets || (h->n_rows > 0 && (!t || !flux || !err || !band))) { g_err = "lcfe_csv_fill: null array"; return 1; }
    const size_t n_obj = h->ids.size();
    offsets[0] = 0;
    for (size_t k = 0; k < n_obj; ++k) offsets[k + 1] = offsets[k] + h->counts[k];
    std::vector<int64_t> cursor(offsets, offsets + n_obj);
    size_t row = 0;
    for (const auto& c : h->chunks) {
        const size_t n = c.t.size();
        for (size_t k = 0; k < n; ++k, ++row) {
            const int64_t dst = cursor[(size_t)h->obj_of_row[row]]++;
            t[dst] = c.t[k];
            flux[dst] = c.f[k];
            err[dst] = c.e[k];
            band[dst] = c.band[k];
        }
    }
    if (id_offsets && id_bytes) {
        int64_t pos = 0;
        for (size_t k = 0; k < n_obj; ++k) {
            id_offsets[k] = pos;
            memcpy(id_bytes + pos, h->ids[k].data(), h->ids[k].size());
            pos += (int64_t)h->ids[k].size();
        }
        id_offsets[n_obj] = pos;
    }
    return 0;
}

}  // extern "C"
