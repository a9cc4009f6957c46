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

#ifdef __cplusplus
}
#endif
#endif /* R433_RECORDS_H_ */
