/*
 * fr_zoompath.c -- the deep-zoom zoom-path animation of DeepZoomManager (host side, as in the reference):
 *   playZoomPath        src/deep_zoom_system.cpp:454-460
 *   zoomTo              :462-485
 *   update_animation    :487-531
 *   interpolate_to_keyframe :533-556   (centre linear, zoom in log space, plain t: "could use smoothstep")
 *   DeepZoomPresets     :575-601
 * The reference keeps centre/zoom in its "ArbitraryFloat", which is a double (src/deep_zoom_system.cpp:19-92); so do we.
 * What the reference's update_animation does besides moving the view -- recompute and upload the reference orbit when a
 * keyframe is reached (:505) -- is the caller's next render here (fr_render recomputes the orbit per Deep_Zoom frame);
 * `orbit_dirty` reports the moment.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "fr_internal.h"

struct fr_zoom_path {
    fr_zoom_keyframe* kf;
    int32_t n;
    int32_t current;           /* _current_keyframe */
    float time;                /* _animation_time   */
    int32_t animating;         /* state.zoom_animating */
    float progress;            /* state.zoom_progress  */
};

int fr_zoom_path_create(fr_zoom_path** out)
{
    if (!out) return fr_set_error(FR_ERR_INVALID_ARG, "fr_zoom_path_create: out is NULL");
    *out = (fr_zoom_path*)calloc(1, sizeof(fr_zoom_path));
    if (!*out) return fr_set_error(FR_ERR_NOMEM, "out of memory");
    return FR_OK;
}

void fr_zoom_path_free(fr_zoom_path* z)
{
    if (!z) return;
    free(z->kf);
    free(z);
}

/* playZoomPath, :454-460 */
int fr_zoom_path_play(fr_zoom_path* z, const fr_zoom_keyframe* path, int32_t n)
{
    if (!z || n < 0 || (n > 0 && !path)) return fr_set_error(FR_ERR_INVALID_ARG, "fr_zoom_path_play: bad argument");
    fr_zoom_keyframe* copy = NULL;
    if (n > 0) {
        for (int32_t i = 0; i < n; ++i)
            if (!(path[i].zoom > 0.0) || !isfinite(path[i].zoom) || !isfinite(path[i].center_x) || !isfinite(path[i].center_y))
                return fr_set_error(FR_ERR_INVALID_ARG, "zoom keyframe %d: zoom must be finite and > 0 (it is interpolated in log space), centre finite", i);
        copy = (fr_zoom_keyframe*)malloc((size_t)n * sizeof(*copy));
        if (!copy) return fr_set_error(FR_ERR_NOMEM, "out of memory");
        memcpy(copy, path, (size_t)n * sizeof(*copy));
    }
    free(z->kf);
    z->kf = copy; z->n = n;
    z->current = 0;
    z->time = 0.0f;
    z->animating = n > 0;
    z->progress = 0.0f;
    return FR_OK;
}

/* zoomTo, :462-485: a two-keyframe path from the current view (duration 0) to the target */
int fr_zoom_path_zoom_to(fr_zoom_path* z, const fr_params* current, double target_x, double target_y, double target_zoom,
                         float duration)
{
    if (!z || !current) return fr_set_error(FR_ERR_INVALID_ARG, "fr_zoom_path_zoom_to: NULL argument");
    fr_zoom_keyframe path[2];
    path[0].center_x = current->center_x; path[0].center_y = current->center_y; path[0].zoom = current->zoom; path[0].duration = 0.0f;
    path[1].center_x = target_x; path[1].center_y = target_y; path[1].zoom = target_zoom; path[1].duration = duration;
    return fr_zoom_path_play(z, path, 2);
}

/* interpolate_to_keyframe, :533-556 */
static void interpolate_to(const fr_zoom_path* z, int32_t index, float t, fr_params* state)
{
    if (index <= 0 || index >= z->n) return;                               /* :534 */
    const fr_zoom_keyframe* prev = &z->kf[index - 1];
    const fr_zoom_keyframe* cur = &z->kf[index];
    const double log_prev = log(prev->zoom), log_cur = log(cur->zoom);     /* :549-550 */
    const double log_z = log_prev + (double)t * (log_cur - log_prev);      /* :551 */
    state->center_x = prev->center_x + (double)t * (cur->center_x - prev->center_x);   /* :553 */
    state->center_y = prev->center_y + (double)t * (cur->center_y - prev->center_y);   /* :554 */
    state->zoom = exp(log_z);                                              /* :555 */
}

/* update_animation, :487-531 */
int fr_zoom_path_update(fr_zoom_path* z, float delta_time, fr_params* state, int32_t* animating, float* progress,
                        int32_t* orbit_dirty)
{
    if (!z || !state) return fr_set_error(FR_ERR_INVALID_ARG, "fr_zoom_path_update: NULL argument");
    if (orbit_dirty) *orbit_dirty = 0;
    if (z->n == 0 || z->current >= z->n) {                                 /* :488-491 */
        z->animating = 0;
    } else {
        z->time += delta_time;                                             /* :493 */
        const fr_zoom_keyframe* kf = &z->kf[z->current];
        if (z->time >= kf->duration) {                                     /* :498: reached: move to the keyframe exactly */
            state->center_x = kf->center_x; state->center_y = kf->center_y; state->zoom = kf->zoom;
            z->current++;
            z->time = 0.0f;
            if (orbit_dirty) *orbit_dirty = 1;                             /* :505 compute_reference_orbit() */
            if (z->current >= z->n) { z->animating = 0; z->progress = 1.0f; }   /* :508-511 */
        } else {
            const float t = z->time / kf->duration;                        /* :515 */
            interpolate_to(z, z->current, t, state);
            float total = 0.0f, elapsed = 0.0f;                            /* :519-528 */
            for (int32_t i = 0; i < z->n; ++i) {
                total += z->kf[i].duration;
                if (i < z->current) elapsed += z->kf[i].duration;
            }
            elapsed += z->time;
            z->progress = total > 0.0f ? elapsed / total : 1.0f;
        }
    }
    if (animating) *animating = z->animating;
    if (progress) *progress = z->progress;
    return FR_OK;
}

/* DeepZoomPresets, :575-601 (0 Seahorse, 1 Elephant, 2 Mini Mandelbrot: the order of src/vk_engine.cpp:963-965) */
int fr_zoom_preset(int32_t which, fr_zoom_keyframe* out)
{
    if (!out) return fr_set_error(FR_ERR_INVALID_ARG, "fr_zoom_preset: out is NULL");
    switch (which) {
    case 0: out->center_x = -0.743643887037151; out->center_y = 0.13182590420533; out->zoom = 1e-6; out->duration = 5.0f; break;
    case 1: out->center_x = -0.7453526; out->center_y = 0.1133189; out->zoom = 1e-8; out->duration = 7.0f; break;
    case 2: out->center_x = -0.74364990; out->center_y = 0.13188204; out->zoom = 1e-10; out->duration = 10.0f; break;
    default: return fr_set_error(FR_ERR_INVALID_ARG, "zoom preset %d: 0 Seahorse, 1 Elephant, 2 Mini Mandelbrot", which);
    }
    return FR_OK;
}
