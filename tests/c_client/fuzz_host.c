#include "fractalrenderer_amd.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
/* Host-side robustness under AddressSanitizer + UBSan (built by tests/test_host.py): feeds truncated and
 * corrupted variants of the sample .franim to the parser, interpolates and re-saves what parses, and pushes
 * multi-band 8- and 16-bit images through the PNG writer.  usage: fuzz_host <sample.franim> <scratch dir> */
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); rewind(f);
    char* buf = malloc(n + 1); if (fread(buf, 1, n, f) != (size_t)n) return 2; fclose(f); buf[n] = 0;
    int ok = 0, bad = 0;
    { fr_anim* a = NULL; if (fr_anim_parse(buf, (size_t)n, &a) != FR_OK) return 6; fr_anim_free(a); }
    for (long cut = 0; cut <= n; cut += 7) {
        fr_anim* a = NULL;
        if (fr_anim_parse(buf, (size_t)cut, &a) == FR_OK) { ok++; fr_anim_free(a); } else bad++;
    }
    unsigned s = 12345;
    for (int t = 0; t < 400; ++t) {
        char* c = malloc(n + 1); memcpy(c, buf, n + 1);
        if (t & 1) {                     /* arbitrary bytes: almost always rejected */
            for (int k = 0; k < 8; ++k) { s = s * 1103515245u + 12345u; c[(s >> 8) % n] = (char)(s >> 24); }
        } else {                         /* digits replaced by digits: still JSON, different numbers */
            for (int k = 0; k < 64; ++k) {
                s = s * 1103515245u + 12345u;
                const long at = (long)((s >> 8) % n);
                if (c[at] >= '0' && c[at] <= '9') c[at] = (char)('0' + (s >> 24) % 10);
            }
        }
        fr_anim* a = NULL;
        if (fr_anim_parse(c, (size_t)n, &a) == FR_OK) {
            ok++; fr_params base, at; fr_params_default(&base);
            for (float tt = -1.0f; tt < 30.0f; tt += 3.7f) fr_anim_state_at(a, tt, &base, &at);
            char tmp[512]; snprintf(tmp, sizeof tmp, "%s/rt.franim", argv[2]); fr_anim_save(a, tmp);
            fr_anim_free(a);
        } else bad++;
        free(c);
    }
    char p1[512], p2[512];
    snprintf(p1, sizeof p1, "%s/big.png", argv[2]); snprintf(p2, sizeof p2, "%s/big16.png", argv[2]);
    static unsigned char img[300 * 4001 * 3];
    for (size_t i = 0; i < sizeof img; ++i) img[i] = (unsigned char)(i * 2654435761u >> 13);
    if (fr_write_png(p1, 4001, 300, 8, img, NULL, 0, 0) != FR_OK) return 3;
    if (fr_write_png(p2, 2000, 300, 16, img, NULL, 0, 1) != FR_OK) return 4;
    {   /* zoom paths: random keyframe lists walked with random time steps, replayed, emptied */
        fr_zoom_path* z = NULL; if (fr_zoom_path_create(&z) != FR_OK) return 7;
        fr_params st; fr_params_default(&st);
        for (int t = 0; t < 200; ++t) {
            fr_zoom_keyframe kf[6]; int nk = (int)((s = s * 1103515245u + 12345u) >> 28) % 7;
            for (int k = 0; k < nk && k < 6; ++k) {
                s = s * 1103515245u + 12345u; kf[k].center_x = (double)(s >> 8) / 1e6 - 8.0;
                s = s * 1103515245u + 12345u; kf[k].center_y = (double)(s >> 8) / 1e6 - 8.0;
                s = s * 1103515245u + 12345u; kf[k].zoom = 1e-12 * (double)((s >> 8) + 1);
                s = s * 1103515245u + 12345u; kf[k].duration = (float)((s >> 24) % 5);
            }
            if (fr_zoom_path_play(z, kf, nk > 6 ? 6 : nk) != FR_OK) return 8;
            for (int u = 0; u < 40; ++u) {
                int anim = 0, dirty = 0; float prog = 0.0f;
                s = s * 1103515245u + 12345u;
                if (fr_zoom_path_update(z, (float)((s >> 20) % 300) / 100.0f, &st, &anim, &prog, &dirty) != FR_OK) return 9;
                if (!(st.zoom > 0.0) || prog < 0.0f || prog > 1.0f) return 10;
            }
            if ((t & 15) == 0) fr_zoom_path_zoom_to(z, &st, 0.25, -0.5, 1e-9, 3.0f);
        }
        fr_zoom_path_free(z);
    }
    free(buf);
    printf("parsed ok %d, rejected %d\n", ok, bad);
    return (argc >= 3 && ok > 0 && bad > 0) ? 0 : 5;
}
