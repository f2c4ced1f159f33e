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
#include <pthread.h>
#include <unistd.h>
#include <signal.h>
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

/* ---- parallel deflate -----------------------------------------------------------------------------------
 * After the render takes < 1 ms the deflate IS the frame time of an animation export (SURVEY.md section 8 f3:
 * an 8192x8192 RGB8 frame is 201 MB through a single-threaded zlib).  The image is cut into bands of ~1 MiB of
 * filtered rows; worker threads compress the bands independently as RAW deflate, every band but the last ending
 * on a sync flush (byte aligned, BFINAL = 0), the last one finishing the stream -- concatenated they are ONE valid
 * deflate stream (what pigz does).  Each band becomes its own IDAT chunk (consecutive IDAT chunks concatenate),
 * the zlib header rides in front of the first, the Adler-32 of the whole (combined from the bands') in a final
 * 4-byte one.  The band partition depends on the image size only, never on the thread count, so the file is
 * byte-identical on every machine. */
typedef struct png_band {
    uint32_t y0, y1;
    uint8_t* out;          /* 4 B length + "IDAT" + [2 B zlib header] + deflate + 4 B CRC, ready to write */
    size_t out_len;
    uLong adler;           /* of the band's filtered bytes */
    size_t raw_len;
    int ok;
} png_band;

typedef struct png_job {
    const void* rgb;
    uint32_t width;
    int bit_depth, level;
    size_t row;            /* bytes of one row of samples */
    png_band* bands;
    uint32_t nbands;
    volatile uint32_t next;
} png_job;

static void png_compress_band(png_job* j, uint32_t b)
{
    png_band* d = &j->bands[b];
    const size_t rows = d->y1 - d->y0, raw_len = (j->row + 1) * rows;
    d->ok = 0; d->out = NULL; d->raw_len = raw_len;
    uint8_t* raw = (uint8_t*)malloc(raw_len);
    if (!raw) return;
    for (uint32_t y = d->y0; y < d->y1; ++y) {
        uint8_t* dst = raw + (j->row + 1) * (y - d->y0);
        *dst++ = 0;                                               /* filter type 0 (None) */
        if (j->bit_depth == 8) memcpy(dst, (const uint8_t*)j->rgb + j->row * y, j->row);
        else {                                                    /* PNG samples are big-endian (png_set_swap, :2198) */
            const uint16_t* s = (const uint16_t*)j->rgb + (size_t)j->width * 3 * y;
            for (size_t k = 0; k < (size_t)j->width * 3; ++k) { dst[2 * k] = (uint8_t)(s[k] >> 8); dst[2 * k + 1] = (uint8_t)s[k]; }
        }
    }
    {   /* adler32 takes a uInt length: feed it in pieces */
        uLong a = adler32(0L, Z_NULL, 0);
        size_t off = 0;
        while (off < raw_len) { const size_t n = raw_len - off > (1u << 30) ? (1u << 30) : raw_len - off; a = adler32(a, raw + off, (uInt)n); off += n; }
        d->adler = a;
    }
    const int first = b == 0, last = b + 1 == j->nbands;
    const size_t bound = compressBound((uLong)raw_len) + 64;
    uint8_t* out = (uint8_t*)malloc(8 + 2 + bound + 4);
    if (!out) { free(raw); return; }
    uint8_t* p = out + 8;
    if (first) { *p++ = 0x78; *p++ = j->level >= 7 ? 0xDA : 0x9C; }   /* zlib header: deflate, 32 KiB window */
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, j->level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { free(raw); free(out); return; }
    size_t in_off = 0, produced = 0;
    int rc = Z_OK;
    for (;;) {                                                    /* avail_in / avail_out are uInt: feed in pieces */
        if (zs.avail_in == 0 && in_off < raw_len) {
            const size_t n = raw_len - in_off > (1u << 30) ? (1u << 30) : raw_len - in_off;
            zs.next_in = raw + in_off; zs.avail_in = (uInt)n; in_off += n;
        }
        const size_t room = bound - produced;
        zs.next_out = p + produced; zs.avail_out = (uInt)(room > (1u << 30) ? (1u << 30) : room);
        const uInt before = zs.avail_out;
        const int flush = in_off < raw_len ? Z_NO_FLUSH : (last ? Z_FINISH : Z_SYNC_FLUSH);
        rc = deflate(&zs, flush);
        produced += before - zs.avail_out;
        if (rc == Z_STREAM_END) break;
        if (rc != Z_OK && rc != Z_BUF_ERROR) break;
        if (flush == Z_SYNC_FLUSH && zs.avail_in == 0 && zs.avail_out != 0) break;   /* flushed completely */
        if (produced >= bound) { rc = Z_BUF_ERROR; break; }
    }
    deflateEnd(&zs);
    free(raw);
    if (!((last && rc == Z_STREAM_END) || (!last && rc == Z_OK))) { free(out); return; }
    const size_t data_len = (size_t)(p - (out + 8)) + produced;
    if (data_len > 0x7FFFFFFFu) { free(out); return; }
    put_be32(out, (uint32_t)data_len);
    memcpy(out + 4, "IDAT", 4);
    uLong crc = crc32(0L, Z_NULL, 0);
    {
        size_t off = 4;
        const size_t endp = 8 + data_len;
        while (off < endp) { const size_t n = endp - off > (1u << 30) ? (1u << 30) : endp - off; crc = crc32(crc, out + off, (uInt)n); off += n; }
    }
    put_be32(out + 8 + data_len, (uint32_t)crc);
    d->out = out; d->out_len = 8 + data_len + 4; d->ok = 1;
}

static void* png_worker(void* arg)
{
    png_job* j = (png_job*)arg;
    for (;;) {
        const uint32_t b = __atomic_fetch_add(&j->next, 1u, __ATOMIC_RELAXED);
        if (b >= j->nbands) break;
        png_compress_band(j, b);
    }
    return NULL;
}

static int png_threads(uint32_t nbands)
{
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    const char* e = getenv("FR_PNG_THREADS");
    if (e && atoi(e) > 0) n = atoi(e);
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    if ((uint32_t)n > nbands) n = (long)nbands;
    return (int)n;
}

int fr_write_png(const char* path, uint32_t width, uint32_t height, int32_t bit_depth, const void* rgb,
                 const fr_png_text* texts, int32_t ntexts, int32_t print_metadata)
{
    if (!path || !rgb || width == 0 || height == 0 || (bit_depth != 8 && bit_depth != 16) || ntexts < 0 || (ntexts && !texts))
        return fr_set_error(FR_ERR_INVALID_ARG, "fr_write_png: bad argument");
    const size_t bpc = (size_t)bit_depth / 8, row = (size_t)width * 3 * bpc;

    /* bands of ~1 MiB of filtered rows (a function of the image size only) */
    uint32_t band_rows = (uint32_t)(((size_t)1 << 20) / (row + 1));
    if (band_rows < 1) band_rows = 1;
    const uint32_t nbands = (height + band_rows - 1) / band_rows;
    png_job job;
    memset(&job, 0, sizeof(job));
    job.rgb = rgb; job.width = width; job.bit_depth = bit_depth; job.row = row;
    job.level = bit_depth == 16 ? 9 : 6;                          /* png_set_compression_level(9), :2132 */
    if (bit_depth == 8) {
        /* 8-bit animation frames: the pixels are what must equal the reference's stbi_write_png output, not the deflate
         * stream.  FR_PNG_LEVEL = 1..9 picks the zlib level (default 6): an animation export is bound by this deflate
         * (profiles/r04_anim_sweep.txt) */
        const char* e = getenv("FR_PNG_LEVEL");
        if (e && atoi(e) >= 1 && atoi(e) <= 9) job.level = atoi(e);
    }
    job.nbands = nbands;
    job.bands = (png_band*)calloc(nbands, sizeof(png_band));
    if (!job.bands) return fr_set_error(FR_ERR_NOMEM, "out of memory");
    for (uint32_t b = 0; b < nbands; ++b) {
        job.bands[b].y0 = b * band_rows;
        job.bands[b].y1 = (b + 1) * band_rows < height ? (b + 1) * band_rows : height;
    }
    const int nthreads = png_threads(nbands);
    pthread_t tid[64];
    int started = 0;
    for (int t = 1; t < nthreads; ++t)
        if (pthread_create(&tid[started], NULL, png_worker, &job) == 0) ++started;
    png_worker(&job);                                             /* the calling thread works too */
    for (int t = 0; t < started; ++t) pthread_join(tid[t], NULL);
    int bands_ok = 1;
    uLong adler = adler32(0L, Z_NULL, 0);
    for (uint32_t b = 0; b < nbands; ++b) {
        if (!job.bands[b].ok) { bands_ok = 0; continue; }
        adler = b == 0 ? job.bands[b].adler : adler32_combine(adler, job.bands[b].adler, (z_off_t)job.bands[b].raw_len);
    }
    if (!bands_ok) {
        for (uint32_t b = 0; b < nbands; ++b) free(job.bands[b].out);
        free(job.bands);
        return fr_set_error(FR_ERR_IO, "deflate failed");
    }

    FILE* f = fopen(path, "wb");
    if (!f) {
        for (uint32_t b = 0; b < nbands; ++b) free(job.bands[b].out);
        free(job.bands);
        return fr_set_error(FR_ERR_IO, "cannot open '%s' for writing: %s", path, strerror(errno));
    }
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
    for (uint32_t b = 0; b < nbands; ++b) {
        if (ok) ok = fwrite(job.bands[b].out, 1, job.bands[b].out_len, f) == job.bands[b].out_len;
        free(job.bands[b].out);
    }
    free(job.bands);
    uint8_t ad[4];
    put_be32(ad, (uint32_t)adler);                                /* the zlib trailer, in an IDAT chunk of its own */
    ok = ok && write_chunk(f, "IDAT", ad, 4);
    ok = ok && write_chunk(f, "IEND", NULL, 0);
    if (fclose(f) != 0) ok = 0;
    if (!ok) return fr_set_error(FR_ERR_IO, "writing '%s' failed", path);
    return FR_OK;
}

/* one frame of packed RGB24 to a file descriptor (e.g. the stdin pipe of
 * `ffmpeg -f rawvideo -pix_fmt rgb24 -s WxH -r FPS -i - ...`), retrying short writes.
 * An encoder that has died must come back as FR_ERR_IO, not as the SIGPIPE that would end the caller's process: the signal is
 * blocked in the writing thread for the duration of the call, and one that the write raised is taken off the thread's
 * pending set before the mask is restored (it is thread-directed; a caller that had it blocked or pending already keeps
 * exactly what it had). */
int fr_write_raw_rgb24(int fd, const uint8_t* rgb8, uint32_t width, uint32_t height)
{
    if (fd < 0 || !rgb8 || width == 0 || height == 0) return fr_set_error(FR_ERR_INVALID_ARG, "fr_write_raw_rgb24: bad argument");
    sigset_t pipe_set, old_set, pending;
    sigemptyset(&pipe_set);
    sigaddset(&pipe_set, SIGPIPE);
    sigpending(&pending);
    const int was_pending = sigismember(&pending, SIGPIPE) == 1;
    const int masked = pthread_sigmask(SIG_BLOCK, &pipe_set, &old_set) == 0;
    int st = FR_OK, err = 0;
    size_t left = (size_t)width * height * 3;
    while (left) {
        const ssize_t n = write(fd, rgb8, left);
        if (n < 0) {
            if (errno == EINTR) continue;
            err = errno;
            st = FR_ERR_IO;
            break;
        }
        rgb8 += n; left -= (size_t)n;
    }
    if (masked) {
        if (err == EPIPE && !was_pending) {
            const struct timespec zero = {0, 0};
            while (sigtimedwait(&pipe_set, NULL, &zero) < 0 && errno == EINTR) {}
        }
        (void)pthread_sigmask(SIG_SETMASK, &old_set, NULL);
    }
    if (st != FR_OK) return fr_set_error(FR_ERR_IO, "write to fd %d failed: %s", fd, strerror(err));
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
