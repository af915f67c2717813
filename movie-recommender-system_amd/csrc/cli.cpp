// cli.cpp — native `knncf` entry points with the reference's CLI and JSON-answer surface.
//
//   knncf baseline              == predict.Baseline              (src/main/scala/predict/Baseline.scala)
//   knncf personalized          == predict.Personalized          (predict/Personalized.scala)
//   knncf knn                   == predict.kNN                   (predict/kNN.scala)
//   knncf distributed-baseline  == distributed.DistributedBaseline (distributed/DistributedBaseline.scala)
//   knncf load-check            parses a ratings file like shared.predictions.load and prints the row count
//
// Flags are Scallop's long options of the reference: --train --test --separator --num_measurements
// --json [--master] (+ additive: --k, --device).  Output: the same JSON keys and nesting, 4-space indent,
// printed and (with --json) saved.  Every number comes from libknncf.so through the C ABI
// (include/knncf.h); this file contains no arithmetic of the path besides mean/std of the timings
// (shared/predictions.scala:18-25).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <fstream>
#include <functional>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/knncf.h"

namespace {

struct Ratings {
    std::vector<int32_t> users, items;
    std::vector<double> ratings;
    int64_t size() const { return (int64_t)users.size(); }
};

std::string trim(const std::string& s) {
    size_t b = 0, e = s.size();
    while (b < e && (unsigned char)s[b] <= ' ') ++b;  // String.trim: code points <= U+0020
    while (e > b && (unsigned char)s[e - 1] <= ' ') --e;
    return s.substr(b, e - b);
}

// Scala's s.toInt: optional sign, decimal digits only, must fit an Int
bool parse_int(const std::string& s, int32_t* out) {
    if (s.empty()) return false;
    size_t i = 0;
    bool neg = false;
    if (s[0] == '-' || s[0] == '+') { neg = s[0] == '-'; i = 1; }
    if (i >= s.size()) return false;
    int64_t v = 0;
    for (; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') return false;
        v = v * 10 + (s[i] - '0');
        if (v > 2147483648ll) return false;
    }
    if (neg) v = -v;
    if (v < -2147483648ll || v > 2147483647ll) return false;
    *out = (int32_t)v;
    return true;
}

// load shared/predictions.scala:35-49 through the library's multithreaded parser (csrc/loader.cpp): header and
// non-numeric-first-column lines are dropped silently, a kept line with a bad column 1 or 2 fails loudly
// --cache-dir DIR: the parse result is kept as DIR/<file name>.knncf (knncf_load_file_cached) and read back while the file
// is unchanged
std::string g_cache_dir;

bool load_ratings(const std::string& path, const std::string& sep, Ratings* out, std::string* err) {
    knncf_ratings r;
    char msg[512] = {0};
    std::string cache;
    if (!g_cache_dir.empty()) {
        const size_t slash = path.find_last_of('/');
        cache = g_cache_dir + "/" + (slash == std::string::npos ? path : path.substr(slash + 1)) + ".knncf";
    }
    int hit = 0;
    if (knncf_load_file_cached(path.c_str(), sep.c_str(), 0, cache.empty() ? nullptr : cache.c_str(), &r, &hit, msg, (int)sizeof msg) != KNNCF_OK) {
        *err = msg;
        return false;
    }
    out->users.assign(r.users, r.users + r.n);
    out->items.assign(r.items, r.items + r.n);
    out->ratings.assign(r.ratings, r.ratings + r.n);
    knncf_free_ratings(&r);
    return true;
}

// ---- tiny ordered JSON writer (ujson.write(obj, 4) look) --------------------------------------------
struct Json {
    enum Kind { NUM, STR, OBJ, ARR, NUL } kind = NUL;
    double num = 0;
    std::string str;
    std::vector<std::pair<std::string, Json>> obj;
    std::vector<Json> arr;
    static Json Num(double v) { Json j; j.kind = NUM; j.num = v; return j; }
    static Json Str(const std::string& s) { Json j; j.kind = STR; j.str = s; return j; }
    static Json Obj() { Json j; j.kind = OBJ; return j; }
    static Json Arr() { Json j; j.kind = ARR; return j; }
    Json& set(const std::string& k, const Json& v) { obj.push_back({k, v}); return *this; }
    Json& push(const Json& v) { arr.push_back(v); return *this; }
};

std::string num_repr(double v) {
    if (isnan(v) || isinf(v)) return "null";
    if (v == floor(v) && fabs(v) < 1e15) {  // ujson prints integral doubles without a fraction
        char b[64];
        snprintf(b, sizeof b, "%.0f", v);
        return b;
    }
    for (int p = 15; p <= 17; ++p) {  // shortest representation that round-trips
        char b[64];
        snprintf(b, sizeof b, "%.*g", p, v);
        if (strtod(b, nullptr) == v) return b;
    }
    char b[64];
    snprintf(b, sizeof b, "%.17g", v);
    return b;
}

void write_json(const Json& j, int depth, std::ostringstream& o) {
    std::string pad((size_t)depth * 4, ' '), pad1((size_t)(depth + 1) * 4, ' ');
    switch (j.kind) {
        case Json::NUM: o << num_repr(j.num); break;
        case Json::NUL: o << "null"; break;
        case Json::STR: {
            o << '"';
            for (char c : j.str) {
                if (c == '"' || c == '\\') o << '\\' << c;
                else if (c == '\n') o << "\\n";
                else if (c == '\t') o << "\\t";
                else o << c;
            }
            o << '"';
            break;
        }
        case Json::OBJ:
            o << "{\n";
            for (size_t i = 0; i < j.obj.size(); ++i) {
                o << pad1 << '"' << j.obj[i].first << "\": ";
                write_json(j.obj[i].second, depth + 1, o);
                o << (i + 1 < j.obj.size() ? ",\n" : "\n");
            }
            o << pad << "}";
            break;
        case Json::ARR:
            o << "[\n";
            for (size_t i = 0; i < j.arr.size(); ++i) {
                o << pad1;
                write_json(j.arr[i], depth + 1, o);
                o << (i + 1 < j.arr.size() ? ",\n" : "\n");
            }
            o << pad << "]";
            break;
    }
}

// mean / std shared/predictions.scala:18-25 (population standard deviation)
double mean(const std::vector<double>& s) {
    if (s.empty()) return 0.0;
    double a = s[0];
    for (size_t i = 1; i < s.size(); ++i) a = a + s[i];
    return a / (double)s.size();
}
double stddev(const std::vector<double>& s) {
    if (s.empty()) return 0.0;
    double m = mean(s), a = 0.0;
    for (double x : s) a = a + (m - x) * (m - x);
    return sqrt(a / (double)s.size());
}
Json timing_obj(const std::vector<double>& t) {
    return Json::Obj().set("average (ms)", Json::Num(mean(t))).set("stddev (ms)", Json::Num(stddev(t)));
}

struct Fail {
    std::string msg;
};

void check(knncf_handle* h, int st, const char* what) {
    if (st != KNNCF_OK) throw Fail{std::string(what) + ": " + (h ? knncf_last_error(h) : knncf_status_string(st))};
}

struct Engine {
    knncf_handle* h = nullptr;
    Engine(int device, int k, int sim) {
        knncf_config cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.struct_size = sizeof cfg;
        cfg.device = device;
        cfg.k = k;
        cfg.similarity = sim;
        cfg.shard_count = 1;
        int st = knncf_create(&cfg, &h);
        if (st != KNNCF_OK) throw Fail{std::string("knncf_create: ") + knncf_status_string(st)};
    }
    ~Engine() { knncf_destroy(h); }
    Engine(const Engine&) = delete;
    void fit(const Ratings& r) { check(h, knncf_fit(h, r.users.data(), r.items.data(), r.ratings.data(), r.size()), "fit"); }
    double mae(int pred, const Ratings& t) {
        double m = 0;
        check(h, knncf_mae(h, pred, t.users.data(), t.items.data(), t.ratings.data(), t.size(), &m), "mae");
        return m;
    }
    double predict(int pred, int32_t u, int32_t i) {
        double p = 0;
        check(h, knncf_predict(h, pred, u, i, &p), "predict");
        return p;
    }
};

// timingInMs shared/predictions.scala:11-16 around `f` (closure construction = fit included)
double timed_ms(const std::function<void()>& f) {
    auto a = std::chrono::steady_clock::now();
    f();
    auto b = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(b - a).count();
}

struct Args {
    std::string cmd, train, test, separator = "\t", json, master, data, personal;
    int num_measurements = 0, k = 300, device = 0;
    bool any_size = false;
};

Json meta(const Args& a, bool with_master) {
    Json m = Json::Obj();
    m.set("1.Train", Json::Str(a.train)).set("2.Test", Json::Str(a.test));
    if (with_master) m.set("3.Master", Json::Str(a.master)).set("4.Measurements", Json::Num(a.num_measurements));
    else m.set("3.Measurements", Json::Num(a.num_measurements));
    return m;
}

Json run_baseline(const Args& a, const Ratings& train, const Ratings& test) {  // predict/Baseline.scala:45-124
    const int kinds[4] = {KNNCF_PRED_GLOBAL_AVG, KNNCF_PRED_USER_AVG, KNNCF_PRED_ITEM_AVG, KNNCF_PRED_BASELINE};
    std::vector<double> times[4];
    for (int p = 0; p < 4; ++p)
        for (int m = 0; m < a.num_measurements; ++m)
            times[p].push_back(timed_ms([&] { Engine e(a.device, a.k, KNNCF_SIM_COSINE); e.fit(train); e.mae(kinds[p], test); }));
    Engine e(a.device, a.k, KNNCF_SIM_COSINE);
    e.fit(train);
    double dev = 0;
    check(e.h, knncf_item_avg_dev(e.h, 1, &dev), "item_avg_dev");
    Json out = Json::Obj();
    out.set("Meta", meta(a, false));
    out.set("B.1", Json::Obj()
                       .set("1.GlobalAvg", Json::Num(e.predict(KNNCF_PRED_GLOBAL_AVG, 1, 1)))
                       .set("2.User1Avg", Json::Num(e.predict(KNNCF_PRED_USER_AVG, 1, 1)))
                       .set("3.Item1Avg", Json::Num(e.predict(KNNCF_PRED_ITEM_AVG, 1, 1)))
                       .set("4.Item1AvgDev", Json::Num(dev))
                       .set("5.PredUser1Item1", Json::Num(e.predict(KNNCF_PRED_BASELINE, 1, 1))));
    out.set("B.2", Json::Obj()
                       .set("1.GlobalAvgMAE", Json::Num(e.mae(KNNCF_PRED_GLOBAL_AVG, test)))
                       .set("2.UserAvgMAE", Json::Num(e.mae(KNNCF_PRED_USER_AVG, test)))
                       .set("3.ItemAvgMAE", Json::Num(e.mae(KNNCF_PRED_ITEM_AVG, test)))
                       .set("4.BaselineMAE", Json::Num(e.mae(KNNCF_PRED_BASELINE, test))));
    out.set("B.3", Json::Obj()
                       .set("1.GlobalAvg", timing_obj(times[0]))
                       .set("2.UserAvg", timing_obj(times[1]))
                       .set("3.ItemAvg", timing_obj(times[2]))
                       .set("4.Baseline", timing_obj(times[3])));
    return out;
}

Json run_knn(const Args& a, const Ratings& train, const Ratings& test) {  // predict/kNN.scala:42-87
    std::vector<double> times;
    for (int m = 0; m < a.num_measurements; ++m)
        times.push_back(timed_ms([&] { Engine e(a.device, a.k, KNNCF_SIM_COSINE); e.fit(train); e.mae(KNNCF_PRED_KNN, test); }));
    Engine e(a.device, 10, KNNCF_SIM_COSINE);
    e.fit(train);
    auto ksim = [&](int32_t u, int32_t v) {  // every answer builds fresh closures in the reference (:66-70)
        check(e.h, knncf_reset_neighbors(e.h), "reset");
        double s = 0;
        check(e.h, knncf_knn_similarity(e.h, u, v, &s), "knn_similarity");
        return s;
    };
    Json out = Json::Obj();
    out.set("Meta", meta(a, false));
    Json n1 = Json::Obj();
    n1.set("1.k10u1v1", Json::Num(ksim(1, 1))).set("2.k10u1v864", Json::Num(ksim(1, 864))).set("3.k10u1v886", Json::Num(ksim(1, 886)));
    check(e.h, knncf_reset_neighbors(e.h), "reset");
    n1.set("4.PredUser1Item1", Json::Num(e.predict(KNNCF_PRED_KNN, 1, 1)));
    out.set("N.1", n1);
    Json maes = Json::Arr();
    for (int k : {10, 30, 50, 100, 200, 300, 400, 800, 943}) {
        check(e.h, knncf_set_k(e.h, k), "set_k");
        maes.push(Json::Arr().push(Json::Num(k)).push(Json::Num(e.mae(KNNCF_PRED_KNN, test))));
    }
    out.set("N.2", Json::Obj().set("1.kNN-Mae", maes));
    out.set("N.3", Json::Obj().set("1.kNN", timing_obj(times)));
    return out;
}

Json run_personalized(const Args& a, const Ratings& train, const Ratings& test) {  // predict/Personalized.scala:54-74
    Json out = Json::Obj();
    out.set("Meta", meta(a, false));
    {
        Engine one(a.device, a.k, KNNCF_SIM_ONE);
        one.fit(train);
        out.set("P.1", Json::Obj()
                           .set("1.PredUser1Item1", Json::Num(one.predict(KNNCF_PRED_PERSONALIZED, 1, 1)))
                           .set("2.OnesMAE", Json::Num(one.mae(KNNCF_PRED_PERSONALIZED, test))));
    }
    {
        Engine cosv(a.device, a.k, KNNCF_SIM_COSINE);
        cosv.fit(train);
        double s21 = 0;
        check(cosv.h, knncf_similarity(cosv.h, 2, 1, &s21), "similarity");
        out.set("P.2", Json::Obj()
                           .set("1.AdjustedCosineUser1User2", Json::Num(s21))
                           .set("2.PredUser1Item1", Json::Num(cosv.predict(KNNCF_PRED_PERSONALIZED, 1, 1)))
                           .set("3.AdjustedCosineMAE", Json::Num(cosv.mae(KNNCF_PRED_PERSONALIZED, test))));
    }
    {
        Engine jac(a.device, a.k, KNNCF_SIM_JACCARD);
        jac.fit(train);
        double s12 = 0;
        check(jac.h, knncf_similarity(jac.h, 1, 2, &s12), "similarity");
        out.set("P.3", Json::Obj()
                           .set("1.JaccardUser1User2", Json::Num(s12))
                           .set("2.PredUser1Item1", Json::Num(jac.predict(KNNCF_PRED_PERSONALIZED, 1, 1)))
                           .set("3.JaccardPersonalizedMAE", Json::Num(jac.mae(KNNCF_PRED_PERSONALIZED, test))));
    }
    return out;
}

Json run_distributed(const Args& a, const Ratings& train, const Ratings& test) {  // distributed/DistributedBaseline.scala:45-83
    std::vector<double> times;
    for (int m = 0; m < a.num_measurements; ++m)
        times.push_back(timed_ms([&] { Engine e(a.device, a.k, KNNCF_SIM_COSINE); e.fit(train); e.mae(KNNCF_PRED_BASELINE_RDD, test); }));
    Engine e(a.device, a.k, KNNCF_SIM_COSINE);
    e.fit(train);
    double dev = 0;
    check(e.h, knncf_item_avg_dev_rdd(e.h, 1, &dev), "item_avg_dev_rdd");
    Json out = Json::Obj();
    out.set("Meta", meta(a, true));
    out.set("D.1", Json::Obj()
                       .set("1.GlobalAvg", Json::Num(e.predict(KNNCF_PRED_GLOBAL_AVG, 1, 1)))
                       .set("2.User1Avg", Json::Num(e.predict(KNNCF_PRED_USER_AVG, 1, 0)))
                       .set("3.Item1Avg", Json::Num(e.predict(KNNCF_PRED_ITEM_AVG, 0, 1)))
                       .set("4.Item1AvgDev", Json::Num(dev))
                       .set("5.PredUser1Item1", Json::Num(e.predict(KNNCF_PRED_BASELINE_RDD, 1, 1)))
                       .set("6.Mae", Json::Num(e.mae(KNNCF_PRED_BASELINE_RDD, test))));
    out.set("D.2", Json::Obj().set("1.DistributedBaseline", timing_obj(times)));
    return out;
}

// recommend/Recommender.scala:40-54 through the library's loader (knncf_load_personal): the non-zero ratings of user 944
// and every row's (id, title)
bool load_personal(const std::string& path, Ratings* out, std::vector<std::pair<int32_t, std::string>>* names, std::string* err) {
    knncf_personal p;
    char msg[512] = {0};
    if (knncf_load_personal(path.c_str(), 944, &p, msg, (int)sizeof msg) != KNNCF_OK) { *err = msg; return false; }
    for (int64_t j = 0; j < p.n_rows; ++j) names->push_back({p.row_ids[j], p.row_names[j]});
    for (int64_t j = 0; j < p.ratings.n; ++j) {
        out->users.push_back(p.ratings.users[j]);
        out->items.push_back(p.ratings.items[j]);
        out->ratings.push_back(p.ratings.ratings[j]);
    }
    knncf_free_personal(&p);
    return true;
}

Json run_recommend(const Args& a, const Ratings& data, const Ratings& personal,
                   const std::vector<std::pair<int32_t, std::string>>& names) {  // recommend/Recommender.scala:68-89
    Ratings aug = data;  // data.union(personal): file order, personal rows last
    aug.users.insert(aug.users.end(), personal.users.begin(), personal.users.end());
    aug.items.insert(aug.items.end(), personal.items.begin(), personal.items.end());
    aug.ratings.insert(aug.ratings.end(), personal.ratings.begin(), personal.ratings.end());
    Engine e(a.device, 300, KNNCF_SIM_COSINE);
    e.fit(aug);
    Json out = Json::Obj();
    out.set("Meta", Json::Obj().set("data", Json::Str(a.data)).set("personal", Json::Str(a.personal)));
    out.set("R.1", Json::Obj().set("PredUser1Item1", Json::Num(e.predict(KNNCF_PRED_KNN, 1, 1))));
    check(e.h, knncf_reset_neighbors(e.h), "reset");  // R.2 builds fresh closures (:85-88)
    int32_t ids[3], cnt = 0;
    double preds[3];
    check(e.h, knncf_recommend(e.h, KNNCF_PRED_KNN, 944, 3, ids, preds, &cnt), "recommend");
    Json r2 = Json::Arr();
    for (int32_t j = 0; j < cnt; ++j) {
        std::string name;
        bool found = false;
        for (const auto& kv : names)  // .toMap: the last row with this id wins
            if (kv.first == ids[j]) { name = kv.second; found = true; }
        if (!found) throw Fail{"recommend: no movie name for item " + std::to_string(ids[j]) + " (movieNames(x._1) throws)"};
        r2.push(Json::Arr().push(Json::Num(ids[j])).push(Json::Str(name)).push(Json::Num(preds[j])));
    }
    out.set("R.2", r2);
    return out;
}

int usage() {
    fprintf(stderr,
            "usage: knncf {baseline|personalized|knn|distributed-baseline|load-check} --train FILE --test FILE\n"
            "             [--separator SEP] [--num_measurements N] [--json FILE] [--master M] [--k K] [--device D]\n"
            "             [--cache-dir DIR]   (binary cache of the parsed files, read back while they are unchanged)\n"
            "       knncf recommend --data FILE --personal FILE [--separator SEP] [--json FILE] [--any-size]\n");
    return 2;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) return usage();
    Args a;
    a.cmd = argv[1];
    for (int i = 2; i < argc; ++i) {
        std::string k = argv[i];
        auto need = [&](const char* name) -> std::string {
            if (i + 1 >= argc) { fprintf(stderr, "[knncf] missing value for %s\n", name); exit(2); }
            return argv[++i];
        };
        if (k == "--train") a.train = need("--train");
        else if (k == "--test") a.test = need("--test");
        else if (k == "--separator") a.separator = need("--separator");
        else if (k == "--num_measurements") a.num_measurements = atoi(need("--num_measurements").c_str());
        else if (k == "--json") a.json = need("--json");
        else if (k == "--master") a.master = need("--master");
        else if (k == "--k") a.k = atoi(need("--k").c_str());
        else if (k == "--device") a.device = atoi(need("--device").c_str());
        else if (k == "--data") a.data = need("--data");
        else if (k == "--personal") a.personal = need("--personal");
        else if (k == "--any-size") a.any_size = true;
        else if (k == "--cache-dir") g_cache_dir = need("--cache-dir");
        else { fprintf(stderr, "[knncf] unknown option %s\n", k.c_str()); return usage(); }
    }
    if (a.separator == "\\t") a.separator = "\t";
    if (a.cmd == "recommend") {
        if (a.data.empty() || a.personal.empty()) { fprintf(stderr, "[knncf] --data and --personal are required\n"); return usage(); }
        Ratings data, personal;
        std::vector<std::pair<int32_t, std::string>> names;
        std::string err;
        printf("\n******************************************************\n");
        printf("Loading data from: %s\n", a.data.c_str());
        if (!load_ratings(a.data, a.separator, &data, &err)) { fprintf(stderr, "[knncf] %s\n", err.c_str()); return 1; }
        if (data.size() != 100000 && !a.any_size) {  // assert(data.length == 100000, "Invalid data") :36
            fprintf(stderr, "[knncf] assertion failed: Invalid data (%lld rows; --any-size lifts the reference's check)\n", (long long)data.size());
            return 1;
        }
        printf("Loading personal data from: %s\n", a.personal.c_str());
        if (!load_personal(a.personal, &personal, &names, &err)) { fprintf(stderr, "[knncf] %s\n", err.c_str()); return 1; }
        try {
            std::ostringstream o;
            write_json(run_recommend(a, data, personal, names), 0, o);
            printf("%s\n", o.str().c_str());
            if (!a.json.empty()) {
                printf("Saving answers in: %s\n", a.json.c_str());
                std::ofstream f(a.json);
                f << o.str();
            }
        } catch (const Fail& f) {
            fprintf(stderr, "[knncf] %s\n", f.msg.c_str());
            return 1;
        }
        printf("\n");
        return 0;
    }
    if (a.train.empty() || (a.test.empty() && a.cmd != "load-check")) { fprintf(stderr, "[knncf] --train and --test are required\n"); return usage(); }
    Ratings train, test;
    std::string err;
    printf("\n******************************************************\n");
    printf("Loading training data from: %s\n", a.train.c_str());
    if (!load_ratings(a.train, a.separator, &train, &err)) { fprintf(stderr, "[knncf] %s\n", err.c_str()); return 1; }
    if (a.cmd == "load-check") {
        printf("rows: %lld\n", (long long)train.size());
        if (train.size() > 0)
            printf("first: %d %d %s\nlast: %d %d %s\n", train.users.front(), train.items.front(), num_repr(train.ratings.front()).c_str(),
                   train.users.back(), train.items.back(), num_repr(train.ratings.back()).c_str());
        return 0;
    }
    printf("Loading test data from: %s\n", a.test.c_str());
    if (!load_ratings(a.test, a.separator, &test, &err)) { fprintf(stderr, "[knncf] %s\n", err.c_str()); return 1; }
    try {
        Json out;
        if (a.cmd == "baseline") out = run_baseline(a, train, test);
        else if (a.cmd == "knn") out = run_knn(a, train, test);
        else if (a.cmd == "personalized") out = run_personalized(a, train, test);
        else if (a.cmd == "distributed-baseline") out = run_distributed(a, train, test);
        else return usage();
        std::ostringstream o;
        write_json(out, 0, o);
        printf("%s\n", o.str().c_str());
        if (!a.json.empty()) {
            printf("Saving answers in: %s\n", a.json.c_str());
            std::ofstream f(a.json);
            f << o.str();
        }
    } catch (const Fail& f) {
        fprintf(stderr, "[knncf] %s\n", f.msg.c_str());
        return 1;
    }
    printf("\n");
    return 0;
}
