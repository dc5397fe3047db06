/*
 * hgi_oracle.c -- CPU restatement of the RustyHGI encode/decode hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under rustyhgi_amd/ links, loads or calls
 * this file.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may use it, and there only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED by the reference's own tests: the reference (Rust, nightly,
 * crates.io deps) cannot be built here, ships no golden vectors, and its four
 * lossy unit tests compare the decoded image with itself (src/lib.rs:58-75).
 * What pins this file instead: SURVEY.md Appendix B (values from an
 * independent reading of the same sources), a second independent numpy
 * restatement (oracle/hgi_numpy.py) and the round-trip invariants in tests/.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  The structure is deliberately the reference's: sequential
 * levels, one scalar visit per pixel in traverse_level order, one thread per
 * image.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define HGI_INTERP_LEFTTOP 0
#define HGI_INTERP_CROSSED 1

/* ---- src/quantizator.rs:41-63  Linear::from(QuantizationLevel) ------------ */
/* level: 0 Lossless, 1 Low, 2 Medium, 3 High (src/quantizator.rs:3-8).       */
int hgi_oracle_linear_lut(int level, uint8_t lut[256], uint8_t *max_err)
{
    static const uint8_t errs[4] = {0, 10, 20, 30};      /* quantizator.rs:43-48 */
    if (level < 0 || level > 3) return 1;
    unsigned error = errs[level];
    unsigned scale = 2 * error + 1;                       /* quantizator.rs:50 */
    for (unsigned i = 0; i < 256; ++i) {                  /* quantizator.rs:57-60 */
        unsigned r = (i + error) / scale;                 /* quantizator.rs:52 */
        unsigned v = r * scale;                           /* quantizator.rs:53 */
        lut[i] = (uint8_t)v;                              /* quantizator.rs:54 `as u8` */
    }
    if (max_err) *max_err = (uint8_t)error;
    return 0;
}

/* ---- src/quantizator.rs:26-33  NoOp::quantize ----------------------------- */
void hgi_oracle_noop_lut(uint8_t lut[256])
{
    for (unsigned i = 0; i < 256; ++i) lut[i] = (uint8_t)i;
}

/* ---- src/interpolator.rs:41-55  CrossedValues::prediction ----------------- */
static inline uint8_t crossed_prediction(uint8_t left_top, uint8_t right_top,
                                         uint8_t left_bot, uint8_t right_bot)
{
#define AVERAGE(x, y) (((size_t)(x) + (size_t)(y) + 1) >> 1) /* interpolator.rs:44 */
    size_t left = AVERAGE(left_top, left_bot);            /* :46 */
    size_t right = AVERAGE(right_bot, right_top);         /* :47 */
    size_t top = AVERAGE(right_top, left_top);            /* :48 */
    size_t bot = AVERAGE(right_bot, left_bot);            /* :49 */
#undef AVERAGE
    size_t average = (left + right + top + bot) >> 2;     /* :51 */
    return (uint8_t)average;                              /* :53 */
}

/* ---- src/interpolator.rs:75-82  get_pixel closure (OOB reads as 0) -------- */
static inline uint8_t get_pixel0(const uint8_t *img, uint32_t w, uint32_t h,
                                 uint64_t x, uint64_t y)
{
    return (x < w && y < h) ? img[(size_t)y * w + (size_t)x] : 0;
}

/* ---- src/interpolator.rs:57-90  Crossed::interpolate ---------------------- */
/* `level` is 1-based exactly as in the reference (callers pass level + 1).   */
static inline uint8_t interpolate_crossed(unsigned levels, unsigned level, uint32_t x,
                                          uint32_t y, const uint8_t *img, uint32_t w,
                                          uint32_t h)
{
    uint64_t step = (uint64_t)1 << (levels - level + 1);  /* :67 */
    uint64_t mask = step - 1;                             /* :68 */
    uint64_t x_top = x - (x & mask);                      /* :70 */
    uint64_t y_left = y - (y & mask);                     /* :71 */
    uint64_t x_bot = x_top + step;                        /* :72 */
    uint64_t y_right = y_left + step;                     /* :73 */
    return crossed_prediction(get_pixel0(img, w, h, x_top, y_left),   /* :85 left_top  */
                              get_pixel0(img, w, h, x_top, y_right),  /* :86 right_top */
                              get_pixel0(img, w, h, x_bot, y_left),   /* :87 left_bot  */
                              get_pixel0(img, w, h, x_bot, y_right)); /* :88 right_bot */
}

/* ---- src/interpolator.rs:15-28  LeftTop::interpolate ---------------------- */
static inline uint8_t interpolate_lefttop(unsigned levels, unsigned level, uint32_t x,
                                          uint32_t y, const uint8_t *img, uint32_t w)
{
    uint64_t step = (uint64_t)1 << (levels - level + 1);  /* :19 */
    uint64_t mask = step - 1;                             /* :20 */
    uint64_t x_top = x - (x & mask);                      /* :22 */
    uint64_t y_left = y - (y & mask);                     /* :23 */
    return img[(size_t)y_left * w + (size_t)x_top];       /* :26 */
}

static inline uint8_t interpolate(int interp, unsigned levels, unsigned level, uint32_t x,
                                  uint32_t y, const uint8_t *img, uint32_t w, uint32_t h)
{
    return interp == HGI_INTERP_CROSSED ? interpolate_crossed(levels, level, x, y, img, w, h)
                                        : interpolate_lefttop(levels, level, x, y, img, w);
}

/* ---- src/utils.rs:12-41  traverse_level ----------------------------------- */
/* Same visiting order as the reference; BODY sees `column` and `line`.       */
#define TRAVERSE_LEVEL(level, levels, x1, x2, y1, y2, BODY)                      \
    do {                                                                         \
        unsigned e_ = (levels) - (level);                       /* utils.rs:16 */ \
        uint64_t step_ = (uint64_t)1 << e_;                     /* :17 */        \
        uint64_t substep_ = (uint64_t)1 << (e_ - 1);            /* :18 */        \
        uint64_t start_ = (x1) + substep_;                      /* :19 */        \
        uint64_t line_ = (y1);                                  /* :21 */        \
        while (line_ < (y2)) {                                  /* :22 */        \
            for (uint64_t c_ = start_; c_ < (x2); c_ += step_) { /* :23-27 */    \
                uint32_t column = (uint32_t)c_, line = (uint32_t)line_;          \
                BODY                                                             \
            }                                                                    \
            line_ += substep_;                                  /* :29 */        \
            if (line_ >= (y2)) break;                           /* :30-32 */     \
            for (uint64_t c_ = (x1); c_ < (x2); c_ += substep_) { /* :34-38 */   \
                uint32_t column = (uint32_t)c_, line = (uint32_t)line_;          \
                BODY                                                             \
            }                                                                    \
            line_ += substep_;                                  /* :39 */        \
        }                                                                        \
    } while (0)

/* ---- src/encoder.rs:39-71  Encoder::encode -------------------------------- */
/* `input` is the caller's image; the reference consumes it by value and      */
/* overwrites it in place with the reconstruction (encoder.rs:64).  Here the  */
/* in-place buffer is `rec` (a private copy, or the caller's rec_out).        */
/* fallbacks (optional) counts how often encoder.rs:58-60 fires.              */
int hgi_oracle_encode(const uint8_t *input, uint32_t width, uint32_t height, uint32_t levels,
                      int interp, const uint8_t lut[256], uint8_t *grid, uint8_t *rec_out,
                      uint64_t *fallbacks)
{
    if (levels > 31 || (interp != HGI_INTERP_CROSSED && interp != HGI_INTERP_LEFTTOP)) return 1;
    size_t n = (size_t)width * height;
    if (n == 0) { if (fallbacks) *fallbacks = 0; return 0; }
    uint8_t *rec = rec_out ? rec_out : (uint8_t *)malloc(n);
    if (!rec) return 2;
    if (rec != input) memcpy(rec, input, n);
    uint64_t nfb = 0;

    /* encoder.rs:26-37 initialize_first_level */
    uint64_t step = (uint64_t)1 << levels;                         /* :28 */
    for (uint64_t line = 0; line < height; line += step)           /* :31 */
        for (uint64_t column = 0; column < width; column += step)  /* :32 */
            grid[line * width + column] = rec[line * width + column]; /* :33-34 */

    for (unsigned level = 0; level < levels; ++level) {            /* :45 */
        TRAVERSE_LEVEL(level, levels, 0, width, 0, height, {       /* :67 */
            uint8_t prediction = interpolate(interp, levels, level + 1, column, line, rec,
                                             width, height);       /* :48-50 */
            uint8_t actual_value = rec[(size_t)line * width + column];      /* :52 */
            uint8_t diff = (uint8_t)(actual_value - prediction);   /* :53 wrapping_sub */
            uint8_t quanted_diff = lut[diff];                      /* :54 */
            int overflow = ((unsigned)prediction + quanted_diff) > 255;     /* :56 */
            int overflow_is_expected = ((unsigned)prediction + diff) > 255; /* :57 */
            if (overflow != overflow_is_expected) {                /* :58 */
                quanted_diff = diff;                               /* :59 */
                ++nfb;
            }
            grid[(size_t)line * width + column] = quanted_diff;    /* :62 */
            rec[(size_t)line * width + column] =
                (uint8_t)(prediction + quanted_diff);              /* :63-64 wrapping_add */
        });
    }
    if (!rec_out) free(rec);
    if (fallbacks) *fallbacks = nfb;
    return 0;
}

/* ---- src/decoder.rs:18-46  Decoder::decode -------------------------------- */
int hgi_oracle_decode(const uint8_t *grid, uint32_t width, uint32_t height, uint32_t levels,
                      int interp, uint8_t *image)
{
    if (levels > 31 || (interp != HGI_INTERP_CROSSED && interp != HGI_INTERP_LEFTTOP)) return 1;
    size_t n = (size_t)width * height;
    if (n == 0) return 0;
    memset(image, 0, n);                                            /* :19 GrayImage::new zeroes */
    uint64_t step = (uint64_t)1 << levels;                          /* :22 */
    for (uint64_t line = 0; line < height; line += step)            /* :23 */
        for (uint64_t column = 0; column < width; column += step)   /* :24 */
            image[line * width + column] = grid[line * width + column]; /* :25-26 */

    for (unsigned level = 0; level < levels; ++level) {             /* :30 */
        TRAVERSE_LEVEL(level, levels, 0, width, 0, height, {        /* :43 */
            uint8_t diff = grid[(size_t)line * width + column];     /* :33 */
            uint8_t prediction = interpolate(interp, levels, level + 1, column, line, image,
                                             width, height);        /* :35-37 */
            image[(size_t)line * width + column] = (uint8_t)(prediction + diff); /* :39-40 */
        });
    }
    return 0;
}

/* ---- src/main.rs:84-92,106,111  `hgi test` SD statistic ------------------- */
/* Returns sum of squared differences; *int_mse gets the INTEGER division.    */
uint64_t hgi_oracle_sq_error(const uint8_t *before, const uint8_t *after, size_t n,
                             uint64_t *int_mse, uint32_t *max_abs)
{
    uint64_t sd = 0;
    uint32_t mx = 0;
    for (size_t i = 0; i < n; ++i) {
        int diff = abs((int)before[i] - (int)after[i]);             /* main.rs:89 */
        sd += (uint64_t)diff * diff;                                /* :91 */
        if ((uint32_t)diff > mx) mx = diff;
    }
    if (int_mse) *int_mse = n ? sd / n : 0;                         /* :106 integer divide */
    if (max_abs) *max_abs = mx;
    return sd;
}

/* ---- synthetic inputs (`xy` is benches/bench.rs:26-28) ---- */
#define HGI_SYNTH_XY 0
#define HGI_SYNTH_NOISE 1
#define HGI_SYNTH_RAMP 2

static inline uint64_t mix64(uint64_t z)
{ /* splitmix64 finaliser */
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static inline uint8_t synth_px(int kind, uint64_t seed, uint64_t frame, uint32_t x, uint32_t y)
{
    if (kind == HGI_SYNTH_XY) return (uint8_t)((uint32_t)(x * y) & 0xFF);
    uint8_t nz = (uint8_t)(mix64(seed ^ (frame << 40) ^ ((uint64_t)y << 20) ^ (uint64_t)x) >> 56);
    if (kind == HGI_SYNTH_NOISE) return nz;
    return (uint8_t)((((3u * x + 5u * y) >> 4) + (nz & 0x0F)) & 0xFF);
}

int hgi_oracle_synth(int kind, uint64_t seed, uint64_t frame, uint32_t w, uint32_t h,
                     uint8_t *out)
{
    if (kind < 0 || kind > 2) return 1;
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) out[(size_t)y * w + x] = synth_px(kind, seed, frame, x, y);
    return 0;
}

/* ---- CPU baseline: one thread per image over a batch ---- */
typedef struct {
    const uint8_t *imgs;
    uint8_t *grids, *outs;
    uint32_t w, h, levels;
    int interp;
    const uint8_t *lut;
    size_t first, count;
    double enc_s, dec_s;
} bench_job;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static void *bench_worker(void *arg)
{
    bench_job *j = (bench_job *)arg;
    size_t n = (size_t)j->w * j->h;
    for (size_t f = j->first; f < j->first + j->count; ++f) {
        double t0 = now_s();
        hgi_oracle_encode(j->imgs + f * n, j->w, j->h, j->levels, j->interp, j->lut,
                          j->grids + f * n, NULL, NULL);
        double t1 = now_s();
        hgi_oracle_decode(j->grids + f * n, j->w, j->h, j->levels, j->interp, j->outs + f * n);
        double t2 = now_s();
        j->enc_s += t1 - t0;
        j->dec_s += t2 - t1;
    }
    return NULL;
}

/* Encode then decode `frames` packed frames with `threads` threads (one image */
/* per thread at a time, like the reference's single-threaded per-image loop). */
/* Returns wall seconds for the whole batch; per-direction CPU-seconds summed  */
/* over threads are written to enc_cpu_s / dec_cpu_s when non-NULL.            */
double hgi_oracle_bench_batch(const uint8_t *imgs, uint8_t *grids, uint8_t *outs, uint32_t w,
                              uint32_t h, uint32_t levels, int interp, const uint8_t lut[256],
                              size_t frames, int threads, double *enc_cpu_s, double *dec_cpu_s)
{
    if (threads < 1) threads = 1;
    if ((size_t)threads > frames && frames > 0) threads = (int)frames;
    pthread_t *tid = (pthread_t *)calloc(threads, sizeof(pthread_t));
    bench_job *jobs = (bench_job *)calloc(threads, sizeof(bench_job));
    size_t per = frames / threads, extra = frames % threads, first = 0;
    double t0 = now_s();
    for (int t = 0; t < threads; ++t) {
        bench_job *j = &jobs[t];
        j->imgs = imgs; j->grids = grids; j->outs = outs;
        j->w = w; j->h = h; j->levels = levels; j->interp = interp; j->lut = lut;
        j->first = first;
        j->count = per + ((size_t)t < extra ? 1 : 0);
        first += j->count;
        pthread_create(&tid[t], NULL, bench_worker, j);
    }
    double es = 0, ds = 0;
    for (int t = 0; t < threads; ++t) {
        pthread_join(tid[t], NULL);
        es += jobs[t].enc_s;
        ds += jobs[t].dec_s;
    }
    double wall = now_s() - t0;
    if (enc_cpu_s) *enc_cpu_s = es;
    if (dec_cpu_s) *dec_cpu_s = ds;
    free(tid);
    free(jobs);
    return wall;
}
