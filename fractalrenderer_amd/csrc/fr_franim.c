/*
 * fr_franim.c -- the .franim animation-parameter format and keyframe interpolation,
 * i.e. what produces the per-frame viewport the renderer is called with:
 *   AnimationSystem::load_from_file / save_to_file   src/animation_system.cpp:221-313
 *   AnimationSystem::interpolate                     src/animation_system.cpp:82-181
 *   AnimationSystem::add_keyframe                    src/animation_system.cpp:12-23
 *   AnimationRenderer::start_render frame arithmetic src/animation_renderer.cpp:48,80
 * The reference parses with nlohmann/json (not vendored, version unpinned); the format is
 * plain JSON, parsed here by a small recursive-descent reader.
 */
#include "fr_internal.h"

#include <ctype.h>
#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- a minimal JSON value tree ------------------------------------------------------------------ */
typedef enum { J_NULL, J_BOOL, J_NUM, J_STR, J_ARR, J_OBJ } jtype;

typedef struct jval {
    jtype type;
    int is_int;             /* number had no fraction/exponent and fits int64 */
    long long i;
    double d;
    int b;
    char* s;                /* J_STR */
    struct jval** items;    /* J_ARR / J_OBJ values */
    char** keys;            /* J_OBJ */
    size_t n, cap;
} jval;

typedef struct { const char* p; const char* end; const char* err; } jparser;

static void jfree(jval* v)
{
    if (!v) return;
    for (size_t k = 0; k < v->n; ++k) {
        jfree(v->items[k]);
        if (v->keys) free(v->keys[k]);
    }
    free(v->items); free(v->keys); free(v->s); free(v);
}

static void jskip(jparser* P)
{
    while (P->p < P->end && (*P->p == ' ' || *P->p == '\t' || *P->p == '\n' || *P->p == '\r')) P->p++;
}

static jval* jnew(jtype t)
{
    jval* v = (jval*)calloc(1, sizeof(jval));
    if (v) v->type = t;
    return v;
}

static int jpush(jval* c, char* key, jval* item)
{
    if (c->n == c->cap) {
        size_t nc = c->cap ? c->cap * 2 : 8;
        jval** ni = (jval**)realloc(c->items, nc * sizeof(jval*));
        if (!ni) return 0;
        c->items = ni;
        if (c->type == J_OBJ) {
            char** nk = (char**)realloc(c->keys, nc * sizeof(char*));
            if (!nk) return 0;
            c->keys = nk;
        }
        c->cap = nc;
    }
    c->items[c->n] = item;
    if (c->type == J_OBJ) c->keys[c->n] = key;
    c->n++;
    return 1;
}

static char* jparse_string_raw(jparser* P)
{
    if (P->p >= P->end || *P->p != '"') { P->err = "expected string"; return NULL; }
    P->p++;
    size_t cap = 32, n = 0;
    char* out = (char*)malloc(cap);
    if (!out) { P->err = "out of memory"; return NULL; }
    while (P->p < P->end && *P->p != '"') {
        unsigned char ch = (unsigned char)*P->p++;
        char buf[4]; size_t bl = 1;
        if (ch == '\\') {
            if (P->p >= P->end) break;
            char e = *P->p++;
            switch (e) {
            case 'n': buf[0] = '\n'; break; case 't': buf[0] = '\t'; break;
            case 'r': buf[0] = '\r'; break; case 'b': buf[0] = '\b'; break;
            case 'f': buf[0] = '\f'; break; case '/': buf[0] = '/'; break;
            case '\\': buf[0] = '\\'; break; case '"': buf[0] = '"'; break;
            case 'u': {
                if (P->end - P->p < 4) { free(out); P->err = "bad \\u escape"; return NULL; }
                unsigned cp = 0;
                for (int k = 0; k < 4; ++k) {
                    char h = *P->p++;
                    cp <<= 4;
                    if (h >= '0' && h <= '9') cp |= (unsigned)(h - '0');
                    else if (h >= 'a' && h <= 'f') cp |= (unsigned)(h - 'a' + 10);
                    else if (h >= 'A' && h <= 'F') cp |= (unsigned)(h - 'A' + 10);
                    else { free(out); P->err = "bad \\u escape"; return NULL; }
                }
                if (cp < 0x80) { buf[0] = (char)cp; }
                else if (cp < 0x800) { buf[0] = (char)(0xC0 | (cp >> 6)); buf[1] = (char)(0x80 | (cp & 0x3F)); bl = 2; }
                else { buf[0] = (char)(0xE0 | (cp >> 12)); buf[1] = (char)(0x80 | ((cp >> 6) & 0x3F)); buf[2] = (char)(0x80 | (cp & 0x3F)); bl = 3; }
                break;
            }
            default: free(out); P->err = "bad escape"; return NULL;
            }
        } else {
            buf[0] = (char)ch;
        }
        if (n + bl + 1 > cap) {
            cap *= 2;
            char* no = (char*)realloc(out, cap);
            if (!no) { free(out); P->err = "out of memory"; return NULL; }
            out = no;
        }
        memcpy(out + n, buf, bl); n += bl;
    }
    if (P->p >= P->end) { free(out); P->err = "unterminated string"; return NULL; }
    P->p++;   /* closing quote */
    out[n] = 0;
    return out;
}

static jval* jparse_value(jparser* P, int depth);

static jval* jparse_number(jparser* P)
{
    const char* s = P->p;
    const char* q = s;
    int is_int = 1;
    if (q < P->end && (*q == '-' || *q == '+')) q++;
    while (q < P->end && (isdigit((unsigned char)*q) || *q == '.' || *q == 'e' || *q == 'E' || *q == '-' || *q == '+')) {
        if (*q == '.' || *q == 'e' || *q == 'E') is_int = 0;
        q++;
    }
    if (q == s) { P->err = "expected number"; return NULL; }
    char tmp[64];
    size_t len = (size_t)(q - s);
    if (len >= sizeof tmp) { P->err = "number too long"; return NULL; }
    memcpy(tmp, s, len); tmp[len] = 0;
    char* endp = NULL;
    errno = 0;
    double d = strtod(tmp, &endp);
    if (endp == tmp || *endp != 0) { P->err = "malformed number"; return NULL; }
    jval* v = jnew(J_NUM);
    if (!v) { P->err = "out of memory"; return NULL; }
    v->d = d;
    v->is_int = 0;
    if (is_int) {
        errno = 0;
        long long ll = strtoll(tmp, &endp, 10);
        if (errno == 0 && *endp == 0) { v->is_int = 1; v->i = ll; }
    }
    P->p = q;
    return v;
}

static jval* jparse_value(jparser* P, int depth)
{
    if (depth > 64) { P->err = "nesting too deep"; return NULL; }
    jskip(P);
    if (P->p >= P->end) { P->err = "unexpected end of input"; return NULL; }
    char ch = *P->p;
    if (ch == '{') {
        P->p++;
        jval* o = jnew(J_OBJ);
        if (!o) { P->err = "out of memory"; return NULL; }
        jskip(P);
        if (P->p < P->end && *P->p == '}') { P->p++; return o; }
        for (;;) {
            jskip(P);
            char* key = jparse_string_raw(P);
            if (!key) { jfree(o); return NULL; }
            jskip(P);
            if (P->p >= P->end || *P->p != ':') { free(key); jfree(o); P->err = "expected ':'"; return NULL; }
            P->p++;
            jval* v = jparse_value(P, depth + 1);
            if (!v) { free(key); jfree(o); return NULL; }
            if (!jpush(o, key, v)) { free(key); jfree(v); jfree(o); P->err = "out of memory"; return NULL; }
            jskip(P);
            if (P->p < P->end && *P->p == ',') { P->p++; continue; }
            if (P->p < P->end && *P->p == '}') { P->p++; return o; }
            jfree(o); P->err = "expected ',' or '}'"; return NULL;
        }
    }
    if (ch == '[') {
        P->p++;
        jval* a = jnew(J_ARR);
        if (!a) { P->err = "out of memory"; return NULL; }
        jskip(P);
        if (P->p < P->end && *P->p == ']') { P->p++; return a; }
        for (;;) {
            jval* v = jparse_value(P, depth + 1);
            if (!v) { jfree(a); return NULL; }
            if (!jpush(a, NULL, v)) { jfree(v); jfree(a); P->err = "out of memory"; return NULL; }
            jskip(P);
            if (P->p < P->end && *P->p == ',') { P->p++; continue; }
            if (P->p < P->end && *P->p == ']') { P->p++; return a; }
            jfree(a); P->err = "expected ',' or ']'"; return NULL;
        }
    }
    if (ch == '"') {
        char* s = jparse_string_raw(P);
        if (!s) return NULL;
        jval* v = jnew(J_STR);
        if (!v) { free(s); P->err = "out of memory"; return NULL; }
        v->s = s;
        return v;
    }
    if ((size_t)(P->end - P->p) >= 4 && !strncmp(P->p, "true", 4)) { P->p += 4; jval* v = jnew(J_BOOL); if (v) v->b = 1; return v; }
    if ((size_t)(P->end - P->p) >= 5 && !strncmp(P->p, "false", 5)) { P->p += 5; jval* v = jnew(J_BOOL); if (v) v->b = 0; return v; }
    if ((size_t)(P->end - P->p) >= 4 && !strncmp(P->p, "null", 4)) { P->p += 4; return jnew(J_NULL); }
    return jparse_number(P);
}

static const jval* jget(const jval* o, const char* key)
{
    if (!o || o->type != J_OBJ) return NULL;
    const jval* found = NULL;
    for (size_t k = 0; k < o->n; ++k)
        if (!strcmp(o->keys[k], key)) found = o->items[k];   /* last duplicate wins, as nlohmann */
    return found;
}

/* nlohmann number conversions: get<T>() is a static_cast from the stored int64/double */
static int jnum_double(const jval* v, double* out)
{
    if (!v) return 0;
    if (v->type == J_NUM) { *out = v->is_int ? (double)v->i : v->d; return 1; }
    if (v->type == J_BOOL) { *out = (double)v->b; return 1; }
    return 0;
}
static int jnum_float(const jval* v, float* out)
{
    if (!v) return 0;
    if (v->type == J_NUM) { *out = v->is_int ? (float)v->i : (float)v->d; return 1; }
    if (v->type == J_BOOL) { *out = (float)v->b; return 1; }
    return 0;
}
static int jnum_int(const jval* v, int32_t* out)
{
    if (!v) return 0;
    if (v->type == J_NUM) { *out = v->is_int ? (int32_t)v->i : (int32_t)v->d; return 1; }
    if (v->type == J_BOOL) { *out = v->b; return 1; }
    return 0;
}
static int jbool(const jval* v, int32_t* out)
{
    if (!v) return 0;
    if (v->type == J_BOOL) { *out = v->b; return 1; }
    return 0;   /* nlohmann throws converting a number to bool */
}

/* ---- the animation ------------------------------------------------------------------------------ */
struct fr_anim {
    char* name;
    char* description;
    float duration;
    int32_t loop;
    int32_t target_fps;
    int32_t export_width, export_height;
    fr_keyframe* kf;
    int32_t n, cap;
};

static char* dupstr(const char* s)
{
    size_t n = strlen(s) + 1;
    char* d = (char*)malloc(n);
    if (d) memcpy(d, s, n);
    return d;
}

/* Animation defaults src/animation_system.h:24-35, AnimationSystem ctor src/animation_system.cpp:7-10 */
int fr_anim_create(fr_anim** out)
{
    if (!out) return fr_set_error(FR_ERR_INVALID_ARG, "out is NULL");
    fr_anim* a = (fr_anim*)calloc(1, sizeof(fr_anim));
    if (!a) return fr_set_error(FR_ERR_NOMEM, "out of memory");
    a->name = dupstr(""); a->description = dupstr("");
    a->duration = 10.0f; a->loop = 0; a->target_fps = 60; a->export_width = 1920; a->export_height = 1080;
    *out = a;
    return FR_OK;
}

void fr_anim_free(fr_anim* a)
{
    if (!a) return;
    free(a->name); free(a->description); free(a->kf); free(a);
}

static int anim_push(fr_anim* a, const fr_keyframe* k)
{
    if (a->n == a->cap) {
        int32_t nc = a->cap ? a->cap * 2 : 8;
        fr_keyframe* nk = (fr_keyframe*)realloc(a->kf, (size_t)nc * sizeof(fr_keyframe));
        if (!nk) return 0;
        a->kf = nk; a->cap = nc;
    }
    a->kf[a->n++] = *k;
    return 1;
}

/* AnimationSystem::add_keyframe, src/animation_system.cpp:12-23.  std::sort there is not
 * stable; equal times keep insertion order here. */
int fr_anim_add_keyframe(fr_anim* a, float time, const fr_params* state, int32_t interp_type)
{
    if (!a || !state) return fr_set_error(FR_ERR_INVALID_ARG, "anim/state is NULL");
    fr_keyframe k;
    k.time = time; k.interp_type = interp_type; k.state = *state;
    if (!anim_push(a, &k)) return fr_set_error(FR_ERR_NOMEM, "out of memory");
    for (int32_t i = a->n - 1; i > 0 && a->kf[i].time < a->kf[i - 1].time; --i) {   /* :16-17 */
        fr_keyframe t = a->kf[i]; a->kf[i] = a->kf[i - 1]; a->kf[i - 1] = t;
    }
    if (time > a->duration) a->duration = time + 1.0f;                               /* :20-22 */
    return FR_OK;
}

/* AnimationSystem::load_from_file body, src/animation_system.cpp:278-304 */
int fr_anim_parse(const char* json, size_t len, fr_anim** out)
{
    if (!json || !out) return fr_set_error(FR_ERR_INVALID_ARG, "json/out is NULL");
    *out = NULL;
    jparser P = {json, json + len, NULL};
    /* UTF-8 BOM */
    if (len >= 3 && (unsigned char)json[0] == 0xEF && (unsigned char)json[1] == 0xBB && (unsigned char)json[2] == 0xBF) P.p += 3;
    jval* root = jparse_value(&P, 0);
    if (!root) return fr_set_error(FR_ERR_PARSE, ".franim: %s at byte %ld", P.err ? P.err : "parse error", (long)(P.p - json));
    jskip(&P);
    if (P.p != P.end) { jfree(root); return fr_set_error(FR_ERR_PARSE, ".franim: trailing data at byte %ld", (long)(P.p - json)); }
    if (root->type != J_OBJ) { jfree(root); return fr_set_error(FR_ERR_PARSE, ".franim: top level is not an object"); }

    fr_anim* a = NULL;
    int st = fr_anim_create(&a);
    if (st != FR_OK) { jfree(root); return st; }

#define REQUIRE(cond, what)                                                             \
    do { if (!(cond)) { jfree(root); fr_anim_free(a);                                   \
         return fr_set_error(FR_ERR_PARSE, ".franim: missing or mistyped key '%s'", what); } } while (0)

    const jval* v;
    v = jget(root, "name");          REQUIRE(v && v->type == J_STR, "name");              /* :280 */
    free(a->name); a->name = dupstr(v->s);
    v = jget(root, "description");   REQUIRE(v && v->type == J_STR, "description");       /* :281 */
    free(a->description); a->description = dupstr(v->s);
    REQUIRE(a->name && a->description, "name");
    REQUIRE(jnum_float(jget(root, "duration"), &a->duration), "duration");                /* :282 */
    REQUIRE(jbool(jget(root, "loop"), &a->loop), "loop");                                 /* :283 */
    REQUIRE(jnum_int(jget(root, "target_fps"), &a->target_fps), "target_fps");            /* :284 */
    REQUIRE(jnum_int(jget(root, "export_width"), &a->export_width), "export_width");      /* :285 */
    REQUIRE(jnum_int(jget(root, "export_height"), &a->export_height), "export_height");   /* :286 */
    const jval* kfs = jget(root, "keyframes");
    REQUIRE(kfs && (kfs->type == J_ARR || kfs->type == J_NULL), "keyframes");             /* :289 */

    for (size_t k = 0; kfs->type == J_ARR && k < kfs->n; ++k) {
        const jval* o = kfs->items[k];
        REQUIRE(o && o->type == J_OBJ, "keyframes[]");
        fr_keyframe kf;
        fr_params_default(&kf.state);                                                     /* FractalState state; :290 */
        REQUIRE(jnum_double(jget(o, "center_x"), &kf.state.center_x), "center_x");        /* :291 */
        REQUIRE(jnum_double(jget(o, "center_y"), &kf.state.center_y), "center_y");        /* :292 */
        REQUIRE(jnum_double(jget(o, "zoom"), &kf.state.zoom), "zoom");                    /* :293 */
        REQUIRE(jnum_int(jget(o, "max_iterations"), &kf.state.max_iterations), "max_iterations");   /* :294 */
        REQUIRE(jnum_int(jget(o, "palette_mode"), &kf.state.palette_mode), "palette_mode");         /* :295 */
        REQUIRE(jnum_float(jget(o, "color_offset"), &kf.state.color_offset), "color_offset");       /* :296 */
        REQUIRE(jnum_float(jget(o, "color_scale"), &kf.state.color_scale), "color_scale");          /* :297 */
        REQUIRE(jnum_float(jget(o, "time"), &kf.time), "time");                           /* :300 */
        REQUIRE(jnum_int(jget(o, "interp_type"), &kf.interp_type), "interp_type");        /* :301 */
        /* ":298 // ... load other state fields ..." -- the writer's remaining keys (:246-255),
         * optional: honoured when present */
        (void)jnum_float(jget(o, "color_brightness"), &kf.state.color_brightness);
        (void)jnum_float(jget(o, "color_saturation"), &kf.state.color_saturation);
        (void)jnum_float(jget(o, "color_contrast"), &kf.state.color_contrast);
        (void)jnum_float(jget(o, "bailout"), &kf.state.bailout);
        (void)jnum_int(jget(o, "antialiasing_samples"), &kf.state.antialiasing_samples);
        { int32_t b; if (jnum_int(jget(o, "orbit_trap_enabled"), &b)) kf.state.orbit_trap_enabled = b ? 1 : 0; }
        (void)jnum_float(jget(o, "orbit_trap_radius"), &kf.state.orbit_trap_radius);
        if (!anim_push(a, &kf)) { jfree(root); fr_anim_free(a); return fr_set_error(FR_ERR_NOMEM, "out of memory"); }   /* :303 push_back, no sort */
    }
#undef REQUIRE
    jfree(root);
    *out = a;
    return FR_OK;
}

int fr_anim_load(const char* path, fr_anim** out)
{
    if (!path || !out) return fr_set_error(FR_ERR_INVALID_ARG, "path/out is NULL");
    FILE* f = fopen(path, "rb");
    if (!f) return fr_set_error(FR_ERR_IO, "cannot open '%s': %s", path, strerror(errno));
    if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return fr_set_error(FR_ERR_IO, "cannot seek '%s'", path); }
    long sz = ftell(f);
    if (sz < 0 || sz > (64L << 20)) { fclose(f); return fr_set_error(FR_ERR_IO, "'%s': unreasonable size", path); }
    rewind(f);
    char* buf = (char*)malloc((size_t)sz + 1);
    if (!buf) { fclose(f); return fr_set_error(FR_ERR_NOMEM, "out of memory"); }
    size_t got = fread(buf, 1, (size_t)sz, f);
    fclose(f);
    buf[got] = 0;
    int st = fr_anim_parse(buf, got, out);
    free(buf);
    return st;
}

/* shortest decimal that round-trips the double, with ".0" on integral values (the shape of
 * nlohmann's dump for number_float) */
static void fmt_double(char* out, size_t cap, double d)
{
    if (isfinite(d) && d == floor(d) && fabs(d) < 1e15) { snprintf(out, cap, "%.1f", d); return; }
    for (int prec = 1; prec <= 17; ++prec) {
        snprintf(out, cap, "%.*g", prec, d);
        if (strtod(out, NULL) == d) return;
    }
}

static void write_escaped(FILE* f, const char* s)
{
    fputc('"', f);
    for (; *s; ++s) {
        unsigned char ch = (unsigned char)*s;
        if (ch == '"') fputs("\\\"", f);
        else if (ch == '\\') fputs("\\\\", f);
        else if (ch == '\n') fputs("\\n", f);
        else if (ch == '\t') fputs("\\t", f);
        else if (ch == '\r') fputs("\\r", f);
        else if (ch < 0x20) fprintf(f, "\\u%04x", ch);
        else fputc(ch, f);
    }
    fputc('"', f);
}

/* AnimationSystem::save_to_file, src/animation_system.cpp:221-273; keys in nlohmann's
 * (alphabetical) object order, 4-space indent as dump(4). */
int fr_anim_save(const fr_anim* a, const char* path)
{
    if (!a || !path) return fr_set_error(FR_ERR_INVALID_ARG, "anim/path is NULL");
    FILE* f = fopen(path, "wb");
    if (!f) return fr_set_error(FR_ERR_IO, "cannot open '%s' for writing: %s", path, strerror(errno));
    char b[64];
    fputs("{\n    \"description\": ", f); write_escaped(f, a->description); fputs(",\n", f);
    fmt_double(b, sizeof b, (double)a->duration); fprintf(f, "    \"duration\": %s,\n", b);
    fprintf(f, "    \"export_height\": %d,\n    \"export_width\": %d,\n", a->export_height, a->export_width);
    fputs("    \"keyframes\": [", f);
    for (int32_t k = 0; k < a->n; ++k) {
        const fr_keyframe* kf = &a->kf[k];
        const fr_params* s = &kf->state;
        fputs(k ? ",\n        {\n" : "\n        {\n", f);
#define PUT_D(key, val, comma) do { fmt_double(b, sizeof b, (double)(val)); fprintf(f, "            \"%s\": %s%s\n", key, b, comma); } while (0)
#define PUT_I(key, val, comma) fprintf(f, "            \"%s\": %d%s\n", key, (int)(val), comma)
        PUT_I("antialiasing_samples", s->antialiasing_samples, ",");
        PUT_D("bailout", s->bailout, ",");
        PUT_D("camera_distance", 3.0f, ",");                  /* 3-D camera: not on this path, FractalState default */
        PUT_D("center_x", s->center_x, ",");
        PUT_D("center_y", s->center_y, ",");
        PUT_D("color_brightness", s->color_brightness, ",");
        PUT_D("color_contrast", s->color_contrast, ",");
        PUT_D("color_offset", s->color_offset, ",");
        PUT_D("color_saturation", s->color_saturation, ",");
        PUT_D("color_scale", s->color_scale, ",");
        PUT_I("interp_type", kf->interp_type, ",");
        PUT_D("mandelbulb_power", 8.0f, ",");
        PUT_I("max_iterations", s->max_iterations, ",");
        fprintf(f, "            \"orbit_trap_enabled\": %s,\n", s->orbit_trap_enabled ? "true" : "false");
        PUT_D("orbit_trap_radius", s->orbit_trap_radius, ",");
        PUT_I("palette_mode", s->palette_mode, ",");
        PUT_D("rotation_y", 0.0f, ",");
        PUT_D("time", kf->time, ",");
        PUT_D("zoom", s->zoom, "");
#undef PUT_D
#undef PUT_I
        fputs("        }", f);
    }
    fputs(a->n ? "\n    ],\n" : "],\n", f);
    fprintf(f, "    \"loop\": %s,\n", a->loop ? "true" : "false");
    fputs("    \"name\": ", f); write_escaped(f, a->name); fputs(",\n", f);
    fprintf(f, "    \"target_fps\": %d\n}", a->target_fps);
    if (fclose(f) != 0) return fr_set_error(FR_ERR_IO, "write to '%s' failed", path);
    return FR_OK;
}

int fr_anim_get_info(const fr_anim* a, fr_anim_info* info)
{
    if (!a || !info) return fr_set_error(FR_ERR_INVALID_ARG, "anim/info is NULL");
    info->duration = a->duration; info->loop = a->loop; info->target_fps = a->target_fps;
    info->export_width = a->export_width; info->export_height = a->export_height;
    info->keyframe_count = a->n;
    return FR_OK;
}

int fr_anim_get_keyframe(const fr_anim* a, int32_t index, fr_keyframe* out)
{
    if (!a || !out) return fr_set_error(FR_ERR_INVALID_ARG, "anim/out is NULL");
    if (index < 0 || index >= a->n) return fr_set_error(FR_ERR_INVALID_ARG, "keyframe index %d out of range", index);
    *out = a->kf[index];
    return FR_OK;
}

const char* fr_anim_name(const fr_anim* a) { return a ? a->name : NULL; }
const char* fr_anim_description(const fr_anim* a) { return a ? a->description : NULL; }

/* ---- interpolation -------------------------------------------------------------------------------- */
/* easing, src/animation_system.cpp:199-212 (float arithmetic; std::pow(float,float) is powf) */
static float ease_in_out(float t) { return t < 0.5f ? 2.0f * t * t : 1.0f - powf(-2.0f * t + 2.0f, 2.0f) / 2.0f; }
static float ease_in(float t) { return t * t; }
static float ease_out(float t) { return 1.0f - (1.0f - t) * (1.0f - t); }

static void keep_unanimated(const fr_params* base, fr_params* out)
{
    out->fractal_type = base->fractal_type;
    out->precision = base->precision;
    out->flags = base->flags;
}

/* AnimationSystem::interpolate, src/animation_system.cpp:82-181 */
int fr_anim_state_at(const fr_anim* a, float time, const fr_params* base, fr_params* out)
{
    if (!a || !base || !out) return fr_set_error(FR_ERR_INVALID_ARG, "anim/base/out is NULL");
    if (a->n == 0) { *out = *base; return FR_OK; }                                   /* :83 */
    if (a->n == 1) { *out = a->kf[0].state; keep_unanimated(base, out); return FR_OK; }   /* :84 */

    time = time < 0.0f ? 0.0f : (a->duration < time ? a->duration : time);          /* :87 std::clamp */

    /* find_keyframe_pair, :183-197 */
    int32_t k1 = a->n - 2, k2 = a->n - 1;
    for (int32_t i = 0; i < a->n - 1; ++i)
        if (time >= a->kf[i].time && time <= a->kf[i + 1].time) { k1 = i; k2 = i + 1; break; }
    const fr_keyframe* key1 = &a->kf[k1];
    const fr_keyframe* key2 = &a->kf[k2];

    const float time_diff = key2->time - key1->time;                                 /* :97 */
    if (time_diff < 0.001f) { *out = key1->state; keep_unanimated(base, out); return FR_OK; }   /* :98-101 */

    float t = (time - key1->time) / time_diff;                                       /* :104 */
    switch (key2->interp_type) {                                                     /* :107-122, type of the SECOND key */
    case FR_INTERP_EASE_IN_OUT: t = ease_in_out(t); break;
    case FR_INTERP_EASE_IN: t = ease_in(t); break;
    case FR_INTERP_EASE_OUT: t = ease_out(t); break;
    case FR_INTERP_EXPONENTIAL: t = t * t; break;
    default: break;
    }

    fr_params r;
    fr_params_default(&r);                                                           /* :125 FractalState result; */
    const fr_params* s1 = &key1->state;
    const fr_params* s2 = &key2->state;

    r.center_x = s1->center_x + t * (s2->center_x - s1->center_x);                   /* :128 (float t widened) */
    r.center_y = s1->center_y + t * (s2->center_y - s1->center_y);                   /* :129 */
    if (s1->zoom > 0.0 && s2->zoom > 0.0) {                                          /* :134-138 log-space */
        const double l1 = log(s1->zoom), l2 = log(s2->zoom);
        r.zoom = exp(l1 + t * (l2 - l1));
    } else {
        r.zoom = s1->zoom + t * (s2->zoom - s1->zoom);                               /* :141 */
    }
    r.zoom = 0.000001 > r.zoom ? 0.000001 : r.zoom;                                  /* :145 std::max */

    float iter_t;                                                                    /* :147-156 */
    if (t < 0.33f) iter_t = 0.0f; else if (t < 0.67f) iter_t = 0.5f; else iter_t = 1.0f;
    r.max_iterations = (int32_t)((float)s1->max_iterations +
                                 iter_t * (float)(s2->max_iterations - s1->max_iterations));   /* :159-161 */

    r.color_offset = s1->color_offset + t * (s2->color_offset - s1->color_offset);   /* :163 */
    r.color_scale = s1->color_scale + t * (s2->color_scale - s1->color_scale);       /* :164 */
    r.color_brightness = s1->color_brightness + t * (s2->color_brightness - s1->color_brightness);   /* :165 */
    r.color_saturation = s1->color_saturation + t * (s2->color_saturation - s1->color_saturation);   /* :166 */
    r.color_contrast = s1->color_contrast + t * (s2->color_contrast - s1->color_contrast);           /* :167 */
    r.palette_mode = (t < 0.5f) ? s1->palette_mode : s2->palette_mode;               /* :169 */
    r.bailout = s1->bailout;                                                         /* :175 */
    r.antialiasing_samples = s1->antialiasing_samples;                               /* :176 */
    r.orbit_trap_enabled = s1->orbit_trap_enabled;                                   /* :177 */
    r.orbit_trap_radius = s1->orbit_trap_radius;                                     /* :178 */
    keep_unanimated(base, &r);
    *out = r;
    return FR_OK;
}

/* src/animation_renderer.cpp:48 */
int32_t fr_anim_frame_count(const fr_anim* a)
{
    if (!a) return 0;
    return (int32_t)(a->duration * (float)a->target_fps);
}

/* src/animation_renderer.cpp:80 */
float fr_anim_frame_time(const fr_anim* a, int32_t frame)
{
    if (!a || a->target_fps == 0) return 0.0f;
    return (float)frame / (float)a->target_fps;
}
