// track_json.cpp -- host-side track loader of libppocar.so.
//
// Follows CarEnv.load_track (reference lib/car_env.py:535-567) and the geometry lists that
// CarEnv.reset builds from it (car_env.py:651-676): every x is scaled by 1280 and every y by
// 720 in float64; walls = consecutive segments of `outer_track_points`, then those of
// `inner_track_points`; gates = consecutive point PAIRS of `reward_gates`.  The schema is the
// one track_editor.py writes (track_editor.py:50-56,126-127).  A minimal recursive-descent JSON
// reader is enough for it (objects, arrays, numbers, strings, true/false/null); numbers go
// through strtod, which is correctly rounded like Python's float().
#include <cctype>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "ppocar_internal.h"

namespace {

struct JVal {
    enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
    double num = 0.0;
    bool b = false;
    std::string str;
    std::vector<JVal> arr;
    std::map<std::string, JVal> obj;
};

struct Parser {
    const char* p;
    const char* end;
    bool ok = true;

    void ws() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    }
    bool lit(const char* s) {
        size_t n = strlen(s);
        if ((size_t)(end - p) >= n && memcmp(p, s, n) == 0) {
            p += n;
            return true;
        }
        return false;
    }
    JVal value(int depth) {
        JVal v;
        ws();
        if (p >= end || depth > 64) {
            ok = false;
            return v;
        }
        if (*p == '{') {
            v.kind = JVal::OBJ;
            ++p;
            ws();
            if (p < end && *p == '}') {
                ++p;
                return v;
            }
            while (ok) {
                ws();
                JVal k = string();
                ws();
                if (!ok || p >= end || *p != ':') {
                    ok = false;
                    break;
                }
                ++p;
                v.obj[k.str] = value(depth + 1);
                ws();
                if (p < end && *p == ',') {
                    ++p;
                    continue;
                }
                if (p < end && *p == '}') {
                    ++p;
                    break;
                }
                ok = false;
            }
        } else if (*p == '[') {
            v.kind = JVal::ARR;
            ++p;
            ws();
            if (p < end && *p == ']') {
                ++p;
                return v;
            }
            while (ok) {
                v.arr.push_back(value(depth + 1));
                ws();
                if (p < end && *p == ',') {
                    ++p;
                    continue;
                }
                if (p < end && *p == ']') {
                    ++p;
                    break;
                }
                ok = false;
            }
        } else if (*p == '"') {
            v = string();
        } else if (lit("true")) {
            v.kind = JVal::BOOL;
            v.b = true;
        } else if (lit("false")) {
            v.kind = JVal::BOOL;
        } else if (lit("null")) {
            v.kind = JVal::NUL;
        } else {
            // JSON number grammar; reject what strtod would accept beyond it (hex, inf, nan)
            const char* q = p;
            if (q < end && *q == '-') ++q;
            if (q >= end || !isdigit((unsigned char)*q)) {
                ok = false;
                return v;
            }
            std::string tok;
            q = p;
            while (q < end && (isdigit((unsigned char)*q) || *q == '-' || *q == '+' || *q == '.' || *q == 'e' || *q == 'E'))
                ++q;
            tok.assign(p, q);
            char* e = nullptr;
            errno = 0;
            v.num = strtod(tok.c_str(), &e);
            if (e == tok.c_str() || *e != '\0') {
                ok = false;
                return v;
            }
            v.kind = JVal::NUM;
            p = q;
        }
        return v;
    }
    JVal string() {
        JVal v;
        v.kind = JVal::STR;
        if (p >= end || *p != '"') {
            ok = false;
            return v;
        }
        ++p;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (p + 1 >= end) {
                    ok = false;
                    return v;
                }
                char c = p[1];
                switch (c) {
                    case 'n': v.str += '\n'; break;
                    case 't': v.str += '\t'; break;
                    case 'r': v.str += '\r'; break;
                    case 'b': v.str += '\b'; break;
                    case 'f': v.str += '\f'; break;
                    case 'u': p += 4; v.str += '?'; break;  // keys of this schema are ASCII
                    default: v.str += c;
                }
                p += 2;
            } else {
                v.str += *p++;
            }
        }
        if (p >= end) {
            ok = false;
            return v;
        }
        ++p;
        return v;
    }
};

bool points(const JVal& root, const char* key, std::vector<double>& xs, std::vector<double>& ys) {
    auto it = root.obj.find(key);
    if (it == root.obj.end() || it->second.kind != JVal::ARR) return false;
    for (const JVal& pt : it->second.arr) {
        if (pt.kind != JVal::ARR || pt.arr.size() != 2 || pt.arr[0].kind != JVal::NUM || pt.arr[1].kind != JVal::NUM)
            return false;
        xs.push_back(pt.arr[0].num * 1280);  // car_env.py:549-565 (scale_factor_x = width = 1280)
        ys.push_back(pt.arr[1].num * 720);   //                    (scale_factor_y = height = 720)
    }
    return true;
}

}  // namespace

int pc_internal_parse_track(const char* path, pc_track* t) {
    FILE* f = fopen(path, "rb");
    if (!f) return PC_ERR_IO;
    std::string buf;
    char tmp[4096];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.append(tmp, n);
    fclose(f);
    Parser ps{buf.data(), buf.data() + buf.size()};
    JVal root = ps.value(0);
    ps.ws();
    if (!ps.ok || root.kind != JVal::OBJ || ps.p != ps.end) return PC_ERR_PARSE;

    std::vector<double> ox, oy, ix, iy, gx, gy;
    if (!points(root, "outer_track_points", ox, oy) || !points(root, "inner_track_points", ix, iy) ||
        !points(root, "reward_gates", gx, gy))
        return PC_ERR_PARSE;
    auto ip = root.obj.find("initial_position");
    auto ia = root.obj.find("initial_angle");
    if (ip == root.obj.end() || ip->second.kind != JVal::ARR || ip->second.arr.size() != 2 ||
        ip->second.arr[0].kind != JVal::NUM || ip->second.arr[1].kind != JVal::NUM || ia == root.obj.end() ||
        ia->second.kind != JVal::NUM)
        return PC_ERR_PARSE;
    if (ox.size() < 2 || ix.size() < 2 || gx.size() < 2) return PC_ERR_PARSE;

    t->walls.clear();
    t->gates.clear();
    for (size_t b = 0; b + 1 < ox.size(); ++b)  // car_env.py:653-661 outer first
        t->walls.insert(t->walls.end(), {ox[b], oy[b], ox[b + 1], oy[b + 1]});
    for (size_t b = 0; b + 1 < ix.size(); ++b)  // car_env.py:662-670 then inner
        t->walls.insert(t->walls.end(), {ix[b], iy[b], ix[b + 1], iy[b + 1]});
    for (size_t g = 0; g + 1 < gx.size(); g += 2)  // car_env.py:671-676 zip(pts[::2], pts[1::2])
        t->gates.insert(t->gates.end(), {gx[g], gy[g], gx[g + 1], gy[g + 1]});
    t->start_x = ip->second.arr[0].num * 1280;  // car_env.py:562-565
    t->start_y = ip->second.arr[1].num * 720;
    t->start_rot = ia->second.num;
    return PC_OK;
}
