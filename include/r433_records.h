/* r433_records.h -- wire format of the compact records that cross the
 * device->host boundary (and that the oracle emits for comparison).
 *
 * All fields little-endian, every record 4-byte aligned.  The formats are a
 * lossless serialisation of the two hand-off structs of the reference:
 *   - pulse_data_t   (reference include/pulse_data.h:30-50)  -> r433_pkg_rec
 *   - bitbuffer_t    (reference include/bitbuffer.h:34-40)   -> r433_evt_rec
 * A "package" is what pulse_detect_package() returns (reference
 * src/pulse_detect.c:199); an "event" is one account_event() call of a slicer
 * (reference src/pulse_slicer.c:26-66), i.e. one bitbuffer handed to a
 * decoder's decode_fn.
 */
#ifndef R433_RECORDS_H_
#define R433_RECORDS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define R433_PD_MAX_PULSES 1200 /* reference include/pulse_data.h:21 */
#define R433_BB_ROWS 50         /* reference include/bitbuffer.h:27 */
#define R433_BB_COLS 128        /* reference include/bitbuffer.h:26 */

#define R433_PKG_OOK 1 /* PULSE_DATA_OOK */
#define R433_PKG_FSK 2 /* PULSE_DATA_FSK */
#define R433_RET_FLUSH 0xffffffffu

/* One detected pulse package.  Followed by num_pulses pairs {int32 pulse, int32 gap}
 * (sample counts), i.e. pulse_data_t.pulse[i] / .gap[i] interleaved.
 * total_bytes = 64 + 8*num_pulses. */
typedef struct r433_pkg_rec {
    uint32_t total_bytes; /* size of this record incl. pulse/gap arrays */
    uint32_t stream;      /* index of the capture in the batch */
    uint32_t type;        /* R433_PKG_OOK | R433_PKG_FSK */
    uint32_t num_pulses;
    uint32_t frame;       /* index of the input frame the package was returned in */
    uint32_t ret_pos;     /* data_counter at return, R433_RET_FLUSH if from the flush call */
    uint64_t offset;      /* pulse_data_t.offset: sample index of first pulse */
    uint32_t start_ago;   /* pulse_data_t.start_ago */
    uint32_t end_ago;     /* pulse_data_t.end_ago */
    int32_t ook_low;      /* ook_low_estimate  */
    int32_t ook_high;     /* ook_high_estimate */
    int32_t fsk_f1;       /* fsk_f1_est */
    int32_t fsk_f2;       /* fsk_f2_est */
    uint32_t sample_rate;
    uint32_t reserved;
} r433_pkg_rec;

/* One bitbuffer handed to a decoder.  Followed by num_rows row entries:
 *   r433_row_rec hdr; uint8_t data[(nbytes+3)&~3];
 * Row r's bytes start at bb[r][0] of the reference bitbuffer and may run on
 * into following physical rows (row spill, reference src/bitbuffer.c:39-54).
 * nbytes covers every byte that was ever written for that row. */
typedef struct r433_evt_rec {
    uint32_t total_bytes; /* size of this record incl. all rows */
    uint32_t pkg;         /* index of the package in canonical batch order */
    uint16_t dev;         /* index of the r_device in registration order */
    uint16_t ordinal;     /* n-th account_event of this (pkg, dev) */
    uint16_t num_rows;    /* bitbuffer_t.num_rows */
    uint16_t free_row;    /* bitbuffer_t.free_row */
} r433_evt_rec;

/* With the decoder pre-filter on (r433_batch_probe_prefilter) a record may be a STUB: an r433_evt_rec alone (total_bytes 16)
 * with num_rows = R433_EVT_STUB and the failure code (0..4 = -return value) in free_row.  It stands for a bitbuffer that its
 * decoder provably refuses with that code, where the decoder sits on a later priority level (reference src/r_api.c:442-451):
 * whether the refusal counts in the decoder's statistics depends on what the levels before it decode of the package, which
 * only the replay knows -- it books a stub it reaches like the refusal, without a call. */
#define R433_EVT_STUB 0xffffu

typedef struct r433_row_rec {
    uint16_t bits;   /* bits_per_row[r] */
    uint16_t syncs;  /* syncs_before_row[r] */
    uint16_t nbytes; /* bytes of row data that follow (before padding) */
    uint16_t reserved;
} r433_row_rec;

/* One slicer timing row: what the fan-out needs to know about an r_device
 * (reference include/r_device.h:59-92; fields read by src/pulse_slicer.c). */
typedef struct r433_dev_timing {
    uint32_t modulation; /* enum modulation_types, reference include/r_device.h:24-40 */
    float short_width;   /* us */
    float long_width;    /* us */
    float reset_limit;   /* us */
    float gap_limit;     /* us */
    float sync_width;    /* us */
    float tolerance;     /* us */
    uint32_t priority;
} r433_dev_timing;

/* Pulse analyzer result for one package (reference src/pulse_analyzer.c:20-35 histogram_t, :279-430): the five
 * width histograms as the reference prints them (fused, in first-seen order) and the decoder it guesses. */
#define R433_HIST_BINS 16 /* MAX_HIST_BINS */
typedef struct r433_hist_bin {
    uint32_t count;
    int32_t sum;
    int32_t mean;
    int32_t min;
    int32_t max;
} r433_hist_bin;

typedef struct r433_histogram {
    uint32_t bins_count;
    r433_hist_bin bins[R433_HIST_BINS];
} r433_histogram;

#define R433_GUESS_NO_PULSES 0
#define R433_GUESS_SINGLE_PULSE 1      /* "Single pulse detected. Probably Frequency Shift Keying or just noise..." */
#define R433_GUESS_UNMODULATED 2       /* "Un-modulated signal. Maybe a preamble..." */
#define R433_GUESS_PPM 3               /* "Pulse Position Modulation with fixed pulse width" */
#define R433_GUESS_PWM_FIXED_GAP 4     /* "Pulse Width Modulation with fixed gap" */
#define R433_GUESS_PWM_FIXED_PERIOD 5  /* "Pulse Width Modulation with fixed period" */
#define R433_GUESS_MANCHESTER 6        /* "Manchester coding" */
#define R433_GUESS_PWM_MULTI 7         /* "Pulse Width Modulation with multiple packets" */
#define R433_GUESS_NRZ 8               /* "Non Return to Zero coding (Pulse Code)" */
#define R433_GUESS_PWM_SYNC 9          /* "Pulse Width Modulation with sync/delimiter" */
#define R433_GUESS_NO_CLUE 10          /* "No clue..." */

typedef struct r433_analysis {
    uint32_t num_pulses;
    int32_t total_period; /* samples from the first pulse to the end of the last one */
    uint32_t guess;       /* R433_GUESS_* */
    uint32_t reserved;
    r433_histogram pulses;
    r433_histogram gaps;       /* without the last gap */
    r433_histogram periods_pg; /* pulse + following gap, without the last */
    r433_histogram periods_gp; /* preceding gap + pulse */
    r433_histogram timings;    /* pulses and gaps together */
    r433_dev_timing device;    /* the guessed decoder as pulse_analyzer leaves it in its r_device; modulation 0: none */
} r433_analysis;

#ifdef __cplusplus
}
#endif
#endif /* R433_RECORDS_H_ */
