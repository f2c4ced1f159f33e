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
    free(buf);
    printf("parsed ok %d, rejected %d\n", ok, bad);
    return (argc >= 3 && ok > 0 && bad > 0) ? 0 : 5;
}
