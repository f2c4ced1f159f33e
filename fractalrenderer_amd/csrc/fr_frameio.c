/*
 * fr_frameio.c -- frame output of the offline path: PNG files and raw RGB24 for an encoder pipe.
 *   8-bit  : VulkanEngine::render_animation_frame writes RGB8 with stbi_write_png, src/vk_engine.cpp:1374-1381
 *   16-bit : export_print_quality writes RGB16 with libpng + gAMA/sRGB/pHYs/tEXt/tIME, src/vk_engine.cpp:2114-2208
 *   raw    : VideoEncoder feeds ffmpeg (src/video_encoder.cpp:195-224); a raw rgb24 pipe needs no PNG round trip
 * The encoder is written here on top of zlib's deflate (stb_image_write / libpng are not in this image);
 * pixel data are what matter for parity, not the compressed byte stream.
 */
#define _POSIX_C_SOURCE 200809L
#include "fr_internal.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>

static void put_be32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }

static int write_chunk(FILE* f, const char type[4], const uint8_t* data, uint32_t len)
{
    uint8_t hdr[8];
    put_be32(hdr, len);
    memcpy(hdr + 4, type, 4);
    uLong crc = crc32(0L, hdr + 4, 4);
    if (len) crc = crc32(crc, data, len);
    uint8_t tail[4];
    put_be32(tail, (uint32_t)crc);
    if (fwrite(hdr, 1, 8, f) != 8) return 0;
    if (len && fwrite(data, 1, len, f) != len) return 0;
    return fwrite(tail, 1, 4, f) == 4;
}

int fr_write_png(const char* path, uint32_t width, uint32_t height, int32_t bit_depth, const void* rgb,
                 const fr_png_text* texts, int32_t ntexts, int32_t print_metadata)
{
    if (!path || !rgb || width == 0 || height == 0 || (bit_depth != 8 && bit_depth != 16) || ntexts < 0 || (ntexts && !texts))
        return fr_set_error(FR_ERR_INVALID_ARG, "fr_write_png: bad argument");
    const size_t bpc = (size_t)bit_depth / 8, row = (size_t)width * 3 * bpc;
    const size_t raw_len = (row + 1) * (size_t)height;
    uint8_t* raw = (uint8_t*)malloc(raw_len);
    if (!raw) return fr_set_error(FR_ERR_NOMEM, "out of memory");
    for (uint32_t y = 0; y < height; ++y) {
        uint8_t* dst = raw + (row + 1) * y;
        *dst++ = 0;                                               /* filter type 0 (None) */
        if (bit_depth == 8) memcpy(dst, (const uint8_t*)rgb + row * y, row);
        else {                                                    /* PNG samples are big-endian (png_set_swap, :2198) */
            const uint16_t* s = (const uint16_t*)rgb + (size_t)width * 3 * y;
            for (size_t k = 0; k < (size_t)width * 3; ++k) { dst[2 * k] = (uint8_t)(s[k] >> 8); dst[2 * k + 1] = (uint8_t)s[k]; }
        }
    }
    uLongf zlen = compressBound((uLong)raw_len);
    uint8_t* z = (uint8_t*)malloc(zlen);
    if (!z) { free(raw); return fr_set_error(FR_ERR_NOMEM, "out of memory"); }
    const int level = bit_depth == 16 ? 9 : 6;                    /* png_set_compression_level(9), :2132 */
    if (compress2(z, &zlen, raw, (uLong)raw_len, level) != Z_OK) { free(raw); free(z); return fr_set_error(FR_ERR_IO, "deflate failed"); }
    free(raw);

    FILE* f = fopen(path, "wb");
    if (!f) { free(z); return fr_set_error(FR_ERR_IO, "cannot open '%s' for writing: %s", path, strerror(errno)); }
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    int ok = fwrite(sig, 1, 8, f) == 8;
    uint8_t ihdr[13];
    put_be32(ihdr, width); put_be32(ihdr + 4, height);
    ihdr[8] = (uint8_t)bit_depth; ihdr[9] = 2 /* RGB */; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;   /* :2135-2142 */
    ok = ok && write_chunk(f, "IHDR", ihdr, 13);
    if (print_metadata) {
        uint8_t b[9];
        put_be32(b, (uint32_t)(1.0 / 2.2 * 100000.0 + 0.5));       /* png_set_gAMA(1/2.2), :2145 */
        ok = ok && write_chunk(f, "gAMA", b, 4);
        b[0] = 0;                                                  /* PNG_sRGB_INTENT_PERCEPTUAL, :2146 */
        ok = ok && write_chunk(f, "sRGB", b, 1);
        const uint32_t ppm = (uint32_t)(300.0 / 0.0254 + 0.5);     /* 300 dpi, :2149-2152 */
        put_be32(b, ppm); put_be32(b + 4, ppm); b[8] = 1;
        ok = ok && write_chunk(f, "pHYs", b, 9);
    }
    for (int32_t k = 0; k < ntexts && ok; ++k) {                   /* png_set_text, PNG_TEXT_COMPRESSION_NONE, :2155-2186 */
        if (!texts[k].key || !texts[k].text) { ok = 0; break; }
        const size_t kl = strlen(texts[k].key), tl = strlen(texts[k].text);
        if (kl < 1 || kl > 79) { ok = 0; break; }
        uint8_t* t = (uint8_t*)malloc(kl + 1 + tl);
        if (!t) { ok = 0; break; }
        memcpy(t, texts[k].key, kl); t[kl] = 0; memcpy(t + kl + 1, texts[k].text, tl);
        ok = write_chunk(f, "tEXt", t, (uint32_t)(kl + 1 + tl));
        free(t);
    }
    if (print_metadata && ok) {                                    /* png_set_tIME, :2189-2192 */
        time_t now = time(NULL);
        struct tm g;
        gmtime_r(&now, &g);
        uint8_t t[7] = {(uint8_t)((g.tm_year + 1900) >> 8), (uint8_t)(g.tm_year + 1900), (uint8_t)(g.tm_mon + 1),
                        (uint8_t)g.tm_mday, (uint8_t)g.tm_hour, (uint8_t)g.tm_min, (uint8_t)g.tm_sec};
        ok = write_chunk(f, "tIME", t, 7);
    }
    ok = ok && write_chunk(f, "IDAT", z, (uint32_t)zlen);
    ok = ok && write_chunk(f, "IEND", NULL, 0);
    free(z);
    if (fclose(f) != 0) ok = 0;
    if (!ok) return fr_set_error(FR_ERR_IO, "writing '%s' failed", path);
    return FR_OK;
}

/* one frame of packed RGB24 to a file descriptor (e.g. the stdin pipe of
 * `ffmpeg -f rawvideo -pix_fmt rgb24 -s WxH -r FPS -i - ...`), retrying short writes */
int fr_write_raw_rgb24(int fd, const uint8_t* rgb8, uint32_t width, uint32_t height)
{
    if (fd < 0 || !rgb8 || width == 0 || height == 0) return fr_set_error(FR_ERR_INVALID_ARG, "fr_write_raw_rgb24: bad argument");
    size_t left = (size_t)width * height * 3;
    while (left) {
        const ssize_t n = write(fd, rgb8, left);
        if (n < 0) {
            if (errno == EINTR) continue;
            return fr_set_error(FR_ERR_IO, "write to fd %d failed: %s", fd, strerror(errno));
        }
        rgb8 += n; left -= (size_t)n;
    }
    return FR_OK;
}

/* "<folder>/frame_%06d.png", src/animation_renderer.cpp:86-88 */
int fr_frame_path(const char* folder, int32_t frame, char* out, size_t cap)
{
    if (!folder || !out || cap == 0 || frame < 0) return fr_set_error(FR_ERR_INVALID_ARG, "fr_frame_path: bad argument");
    const int n = snprintf(out, cap, "%s/frame_%06d.png", folder, frame);
    if (n < 0 || (size_t)n >= cap) return fr_set_error(FR_ERR_INVALID_ARG, "fr_frame_path: buffer too small");
    return FR_OK;
}
