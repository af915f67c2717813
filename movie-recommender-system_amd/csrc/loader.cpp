// loader.cpp — `load` shared/predictions.scala:35-49 as a multithreaded host parser (at ml-25m the
// reference's parse time dwarfs the GPU time).  Semantics of the reference, line by line:
//   * split the line on the separator, trim every column (String.trim: code points <= U+0020);
//   * keep the line iff column 0 parses as an Int (`toInt` :27-33 wraps the NumberFormatException): header lines
//     and garbage are dropped SILENTLY;
//   * columns 1 and 2 of a kept line must parse (the reference throws NumberFormatException /
//     ArrayIndexOutOfBoundsException: this loader fails loudly with the line number); further columns (timestamp)
//     are ignored.
// The file is read once, cut into byte ranges at line boundaries, parsed by `threads` workers into private arrays and
// concatenated in file order (the order is part of the reference's semantics: SURVEY N2/N4).
#include <errno.h>
#include <sys/stat.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/knncf.h"

namespace {

struct Chunk {
    std::vector<int32_t> users, items;
    std::vector<double> ratings;
    int64_t bad_offset = -1;  // byte offset of the first malformed kept line
    const char* bad_what = nullptr;
};

inline bool is_space(unsigned char c) { return c <= ' '; }

// Scala's s.toInt on a trimmed column: optional sign, decimal digits only, must fit an Int
inline bool parse_int(const char* b, const char* e, int32_t* out) {
    while (b < e && is_space((unsigned char)*b)) ++b;
    while (e > b && is_space((unsigned char)e[-1])) --e;
    if (b >= e) return false;
    bool neg = false;
    if (*b == '-' || *b == '+') { neg = *b == '-'; ++b; }
    if (b >= e) return false;
    int64_t v = 0;
    for (; b < e; ++b) {
        if (*b < '0' || *b > '9') return false;
        v = v * 10 + (*b - '0');
        if (v > 2147483648ll) return false;
    }
    if (neg) v = -v;
    if (v < -2147483648ll || v > 2147483647ll) return false;
    *out = (int32_t)v;
    return true;
}

// Scala's s.toDouble on a trimmed column (java.lang.Double.parseDouble): strtod over exactly the column
inline bool parse_double(const char* b, const char* e, double* out) {
    while (b < e && is_space((unsigned char)*b)) ++b;
    while (e > b && is_space((unsigned char)e[-1])) --e;
    if (b >= e) return false;
    char buf[64];
    const size_t len = (size_t)(e - b);
    if (len >= sizeof buf) return false;
    memcpy(buf, b, len);
    buf[len] = '\0';
    char* endp = nullptr;
    const double v = strtod(buf, &endp);
    if (endp != buf + len) return false;
    *out = v;
    return true;
}

void parse_range(const char* data, int64_t begin, int64_t end, const char* sep, size_t sep_len, Chunk* out) {
    int64_t pos = begin;
    while (pos < end) {
        const char* line = data + pos;
        const char* nl = (const char*)memchr(line, '\n', (size_t)(end - pos));
        const char* le = nl ? nl : data + end;
        const int64_t next = nl ? (nl - data) + 1 : end;
        if (le > line && le[-1] == '\r') --le;
        // columns 0, 1, 2 (String.split drops trailing empty strings: an empty 3rd column does not exist)
        const char* cb[3];
        const char* ce[3];
        int ncol = 0;
        const char* p = line;
        while (ncol < 3) {
            const char* q = nullptr;
            if (sep_len > 0 && (size_t)(le - p) >= sep_len) {
                for (const char* s = p; s + sep_len <= le; ++s)
                    if (memcmp(s, sep, sep_len) == 0) { q = s; break; }
            }
            cb[ncol] = p;
            ce[ncol] = q ? q : le;
            ++ncol;
            if (!q) break;
            p = q + sep_len;
        }
        int32_t u;
        if (parse_int(cb[0], ce[0], &u)) {
            // trailing empties dropped: a kept line needs non-empty columns 1 and 2
            int32_t i = 0;
            double r = 0;
            if (ncol < 3 || !parse_int(cb[1], ce[1], &i)) {
                if (out->bad_offset < 0) { out->bad_offset = pos; out->bad_what = "malformed rating row"; }
                return;
            }
            if (!parse_double(cb[2], ce[2], &r)) {
                if (out->bad_offset < 0) { out->bad_offset = pos; out->bad_what = "malformed rating value"; }
                return;
            }
            out->users.push_back(u);
            out->items.push_back(i);
            out->ratings.push_back(r);
        }
        pos = next;
    }
}

void set_err(char* err, int cap, const std::string& msg) {
    if (err && cap > 0) snprintf(err, (size_t)cap, "%s", msg.c_str());
}

}  // namespace

extern "C" {

int knncf_load_file(const char* path, const char* separator, int threads, knncf_ratings* out, char* err, int err_cap) {
    if (!path || !separator || !out) { set_err(err, err_cap, "null argument"); return KNNCF_E_INVALID; }
    out->n = 0;
    out->users = out->items = nullptr;
    out->ratings = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) { set_err(err, err_cap, std::string("cannot open ") + path + ": " + strerror(errno)); return KNNCF_E_INVALID; }
    std::vector<char> data;
    {
        fseek(f, 0, SEEK_END);
        const long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        if (sz > 0) {
            data.resize((size_t)sz);
            const size_t got = fread(data.data(), 1, (size_t)sz, f);
            data.resize(got);
        } else {  // not seekable (a pipe): read to the end
            char buf[1 << 16];
            size_t got;
            while ((got = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + got);
        }
        fclose(f);
    }
    const int64_t size = (int64_t)data.size();
    int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 64) nt = 64;
    if (size < (1 << 20)) nt = 1;
    // byte ranges that start right after a newline
    std::vector<int64_t> cut(nt + 1, size);
    cut[0] = 0;
    for (int t = 1; t < nt; ++t) {
        int64_t p = size * t / nt;
        while (p < size && data[(size_t)p - 1] != '\n') ++p;
        cut[t] = p;
    }
    for (int t = 1; t <= nt; ++t) cut[t] = std::max(cut[t], cut[t - 1]);
    std::vector<Chunk> chunks(nt);
    const size_t sep_len = strlen(separator);
    std::vector<std::thread> workers;
    for (int t = 1; t < nt; ++t)
        workers.emplace_back(parse_range, data.data(), cut[t], cut[t + 1], separator, sep_len, &chunks[t]);
    parse_range(data.data(), cut[0], cut[1], separator, sep_len, &chunks[0]);
    for (auto& w : workers) w.join();
    int64_t total = 0;
    for (int t = 0; t < nt; ++t) {
        if (chunks[t].bad_offset >= 0) {  // the first malformed line in file order (earlier chunks are complete)
            int64_t lineno = 1;
            for (int64_t p = 0; p < chunks[t].bad_offset; ++p) lineno += data[(size_t)p] == '\n';
            set_err(err, err_cap, std::string(path) + ":" + std::to_string(lineno) + ": " + chunks[t].bad_what);
            return KNNCF_E_INVALID;
        }
        total += (int64_t)chunks[t].users.size();
    }
    const size_t cnt = total > 0 ? (size_t)total : 1;
    out->users = (int32_t*)malloc(cnt * sizeof(int32_t));
    out->items = (int32_t*)malloc(cnt * sizeof(int32_t));
    out->ratings = (double*)malloc(cnt * sizeof(double));
    if (!out->users || !out->items || !out->ratings) {
        knncf_free_ratings(out);
        set_err(err, err_cap, "out of host memory");
        return KNNCF_E_NOMEM;
    }
    int64_t at = 0;
    for (int t = 0; t < nt; ++t) {
        const size_t m = chunks[t].users.size();
        if (m) {
            memcpy(out->users + at, chunks[t].users.data(), m * sizeof(int32_t));
            memcpy(out->items + at, chunks[t].items.data(), m * sizeof(int32_t));
            memcpy(out->ratings + at, chunks[t].ratings.data(), m * sizeof(double));
        }
        at += (int64_t)m;
    }
    out->n = total;
    return KNNCF_OK;
}

// ---- binary cache of a parsed ratings file -------------------------------------------------------------
// What is expensive on the host side of a fit is the text parse (seconds at ml-25m against a 6 ms K0 on the GPU); the CSR /
// CSC themselves are rebuilt faster than they could be read back.  So the cache holds the parse's result — the triples in
// FILE ORDER, which the reference's summation orders depend on — stamped with the source file's size and modification time
// and the separator it was split on, and closed by a checksum.
namespace {
constexpr char RATINGS_MAGIC[8] = {'K', 'N', 'C', 'F', 'R', 'A', 'T', '1'};
struct CacheHeader {
    char magic[8];
    int64_t n;
    int64_t src_size;
    int64_t src_mtime_ns;
    uint32_t sep_len;
    char sep[20];
    uint64_t checksum;  // of the three arrays
};

uint64_t fold_words(const void* p, size_t bytes, uint64_t h) {
    const unsigned char* b = (const unsigned char*)p;
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) {
        uint64_t w;
        memcpy(&w, b + i, 8);
        h = (h ^ w) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
    }
    for (; i < bytes; ++i) h = (h ^ b[i]) * 0x100000001B3ull;
    return h;
}
uint64_t ratings_checksum(const knncf_ratings& r) {
    uint64_t h = 0xcbf29ce484222325ull ^ (uint64_t)r.n;
    h = fold_words(r.users, (size_t)r.n * sizeof(int32_t), h);
    h = fold_words(r.items, (size_t)r.n * sizeof(int32_t), h);
    return fold_words(r.ratings, (size_t)r.n * sizeof(double), h);
}
bool stat_source(const char* path, int64_t* size, int64_t* mtime_ns) {
    struct stat st;
    if (stat(path, &st) != 0) return false;
    *size = (int64_t)st.st_size;
    *mtime_ns = (int64_t)st.st_mtim.tv_sec * 1000000000ll + (int64_t)st.st_mtim.tv_nsec;
    return true;
}
bool read_exact(FILE* f, void* p, size_t bytes) { return bytes == 0 || fread(p, 1, bytes, f) == bytes; }

// true: `out` holds the cached triples; false: no usable cache (missing, stale, truncated or corrupt) — never an error
bool read_ratings_cache(const char* cache_path, int64_t src_size, int64_t src_mtime_ns, const char* separator, knncf_ratings* out) {
    FILE* f = fopen(cache_path, "rb");
    if (!f) return false;
    CacheHeader h;
    bool ok = read_exact(f, &h, sizeof h) && memcmp(h.magic, RATINGS_MAGIC, 8) == 0 && h.n >= 0 && h.src_size == src_size &&
              h.src_mtime_ns == src_mtime_ns && h.sep_len == strlen(separator) && h.sep_len <= sizeof h.sep &&
              memcmp(h.sep, separator, h.sep_len) == 0;
    if (ok) {
        // the header's row count is checked against the file's own length before anything is allocated: a corrupt or
        // truncated cache with a huge n must not ask for terabytes (which overcommit grants lazily) — and n stays below the
        // fit's own limit of 2^29 rows, so the byte counts below cannot wrap
        struct stat cst;
        ok = h.n < ((int64_t)1 << 29) && fstat(fileno(f), &cst) == 0 &&
             (int64_t)cst.st_size == (int64_t)sizeof(CacheHeader) + h.n * 16;
    }
    if (ok) {
        const size_t cnt = h.n > 0 ? (size_t)h.n : 1;
        out->users = (int32_t*)malloc(cnt * sizeof(int32_t));
        out->items = (int32_t*)malloc(cnt * sizeof(int32_t));
        out->ratings = (double*)malloc(cnt * sizeof(double));
        out->n = h.n;
        ok = out->users && out->items && out->ratings && read_exact(f, out->users, (size_t)h.n * sizeof(int32_t)) &&
             read_exact(f, out->items, (size_t)h.n * sizeof(int32_t)) && read_exact(f, out->ratings, (size_t)h.n * sizeof(double)) &&
             fgetc(f) == EOF && ratings_checksum(*out) == h.checksum;
        if (!ok) knncf_free_ratings(out);
    }
    fclose(f);
    return ok;
}

// best effort (a cache that cannot be written is not an error of the load): tmp file + rename, so that a reader never sees
// a half-written cache
void write_ratings_cache(const char* cache_path, int64_t src_size, int64_t src_mtime_ns, const char* separator, const knncf_ratings& r) {
    CacheHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, RATINGS_MAGIC, 8);
    h.n = r.n;
    h.src_size = src_size;
    h.src_mtime_ns = src_mtime_ns;
    h.sep_len = (uint32_t)strlen(separator);
    if (h.sep_len > sizeof h.sep) return;
    memcpy(h.sep, separator, h.sep_len);
    h.checksum = ratings_checksum(r);
    const std::string tmp = std::string(cache_path) + ".tmp." + std::to_string((long long)getpid());
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return;
    const size_t n = (size_t)r.n;
    const bool ok = fwrite(&h, 1, sizeof h, f) == sizeof h && (n == 0 || (fwrite(r.users, sizeof(int32_t), n, f) == n &&
                    fwrite(r.items, sizeof(int32_t), n, f) == n && fwrite(r.ratings, sizeof(double), n, f) == n));
    const bool closed = fclose(f) == 0;
    if (!ok || !closed || rename(tmp.c_str(), cache_path) != 0) remove(tmp.c_str());
}
}  // namespace

int knncf_load_file_cached(const char* path, const char* separator, int threads, const char* cache_path, knncf_ratings* out,
                           int* from_cache, char* err, int err_cap) {
    if (from_cache) *from_cache = 0;
    if (!cache_path) return knncf_load_file(path, separator, threads, out, err, err_cap);
    if (!path || !separator || !out) { set_err(err, err_cap, "null argument"); return KNNCF_E_INVALID; }
    out->n = 0;
    out->users = out->items = nullptr;
    out->ratings = nullptr;
    int64_t size = 0, mtime_ns = 0;
    const bool have_stat = stat_source(path, &size, &mtime_ns);  // (a pipe, a missing file: the parser reports what is wrong)
    if (have_stat && read_ratings_cache(cache_path, size, mtime_ns, separator, out)) {
        if (from_cache) *from_cache = 1;
        return KNNCF_OK;
    }
    const int st = knncf_load_file(path, separator, threads, out, err, err_cap);
    if (st == KNNCF_OK && have_stat) write_ratings_cache(cache_path, size, mtime_ns, separator, *out);
    return st;
}

// recommend/Recommender.scala:40-54: the Recommender's personal-ratings file.  Per line: split on "," (Java's
// String.split: trailing EMPTY strings are dropped — before the trim), trim every column; column 0 == "id" is the header
// row -> (0, "header") in the name list and Rating(user, 0, 0.0); fewer than 3 columns -> rating 0.0; otherwise
// cols(2).toDouble.  Ratings equal to 0 are filtered out (:50).  cols(0).toInt / cols(2).toDouble throw in the reference:
// KNNCF_E_INVALID with the line number here.  Every row's (id, cols(1)) goes into the name list (:51-54).
int knncf_load_personal(const char* path, int32_t user, knncf_personal* out, char* err, int err_cap) {
    if (!path || !out) { set_err(err, err_cap, "null argument"); return KNNCF_E_INVALID; }
    memset(out, 0, sizeof *out);
    FILE* f = fopen(path, "rb");
    if (!f) { set_err(err, err_cap, std::string("cannot open ") + path + ": " + strerror(errno)); return KNNCF_E_INVALID; }
    std::string data;
    {
        char buf[1 << 16];
        size_t got;
        while ((got = fread(buf, 1, sizeof buf, f)) > 0) data.append(buf, got);
        fclose(f);
    }
    std::vector<int32_t> ids, r_items;
    std::vector<std::string> names;
    std::vector<double> r_values;
    auto trimmed = [](const std::string& x) {
        size_t b = 0, e = x.size();
        while (b < e && is_space((unsigned char)x[b])) ++b;
        while (e > b && is_space((unsigned char)x[e - 1])) --e;
        return x.substr(b, e - b);
    };
    int64_t lineno = 0;
    size_t pos = 0;
    while (pos < data.size()) {  // textFile: one record per line, no record for the empty tail after the last newline
        size_t nl = data.find('\n', pos);
        std::string line = data.substr(pos, nl == std::string::npos ? std::string::npos : nl - pos);
        pos = nl == std::string::npos ? data.size() : nl + 1;
        ++lineno;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::vector<std::string> cols;
        for (size_t p = 0;;) {
            size_t q = line.find(',', p);
            cols.push_back(line.substr(p, q == std::string::npos ? std::string::npos : q - p));
            if (q == std::string::npos) break;
            p = q + 1;
        }
        while (cols.size() > 1 && cols.back().empty()) cols.pop_back();  // String.split(",") drops trailing empty strings
        if (cols.size() == 1 && cols[0].empty()) cols[0] = "";           // "".split(",") = Array("")
        for (auto& c : cols) c = trimmed(c);
        const std::string where = std::string(path) + ":" + std::to_string(lineno) + ": ";
        if (cols[0] == "id") {
            ids.push_back(0);
            names.push_back("header");
            continue;
        }
        int32_t id;
        if (!parse_int(cols[0].data(), cols[0].data() + cols[0].size(), &id)) {
            set_err(err, err_cap, where + "column 0 is not an Int");
            return KNNCF_E_INVALID;
        }
        if (cols.size() < 2) {  // cols(1) on a one-column row: ArrayIndexOutOfBoundsException
            set_err(err, err_cap, where + "no title column");
            return KNNCF_E_INVALID;
        }
        ids.push_back(id);
        names.push_back(cols[1]);
        if (cols.size() < 3) continue;  // Rating(user, id, 0.0): filtered
        double r;
        if (!parse_double(cols[2].data(), cols[2].data() + cols[2].size(), &r)) {
            set_err(err, err_cap, where + "column 2 is not a Double");
            return KNNCF_E_INVALID;
        }
        if (r != 0) {
            r_items.push_back(id);
            r_values.push_back(r);
        }
    }
    const size_t nr = ids.size(), nq = r_items.size();
    size_t name_bytes = 0;
    for (const auto& nm : names) name_bytes += nm.size() + 1;
    out->row_ids = (int32_t*)malloc((nr ? nr : 1) * sizeof(int32_t));
    out->row_names = (char**)malloc((nr ? nr : 1) * sizeof(char*));
    out->name_storage = (char*)malloc(name_bytes ? name_bytes : 1);
    out->ratings.users = (int32_t*)malloc((nq ? nq : 1) * sizeof(int32_t));
    out->ratings.items = (int32_t*)malloc((nq ? nq : 1) * sizeof(int32_t));
    out->ratings.ratings = (double*)malloc((nq ? nq : 1) * sizeof(double));
    if (!out->row_ids || !out->row_names || !out->name_storage || !out->ratings.users || !out->ratings.items || !out->ratings.ratings) {
        knncf_free_personal(out);
        set_err(err, err_cap, "out of host memory");
        return KNNCF_E_NOMEM;
    }
    char* w = out->name_storage;
    for (size_t j = 0; j < nr; ++j) {
        out->row_ids[j] = ids[j];
        out->row_names[j] = w;
        memcpy(w, names[j].c_str(), names[j].size() + 1);
        w += names[j].size() + 1;
    }
    for (size_t j = 0; j < nq; ++j) {
        out->ratings.users[j] = user;
        out->ratings.items[j] = r_items[j];
        out->ratings.ratings[j] = r_values[j];
    }
    out->n_rows = (int64_t)nr;
    out->ratings.n = (int64_t)nq;
    return KNNCF_OK;
}

void knncf_free_personal(knncf_personal* p) {
    if (!p) return;
    free(p->row_ids);
    free(p->row_names);
    free(p->name_storage);
    knncf_free_ratings(&p->ratings);
    p->row_ids = nullptr;
    p->row_names = nullptr;
    p->name_storage = nullptr;
    p->n_rows = 0;
}

void knncf_free_ratings(knncf_ratings* r) {
    if (!r) return;
    free(r->users);
    free(r->items);
    free(r->ratings);
    r->users = r->items = nullptr;
    r->ratings = nullptr;
    r->n = 0;
}

}  // extern "C"
