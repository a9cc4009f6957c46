/* r433_abi.h -- layout-compatible mirrors of the three reference structs that cross the plugin
 * boundary.  A decoder compiled against the reference headers can be handed these objects
 * unchanged (and vice versa): same field order, types, sizes and offsets on LP64.
 *
 *   r433_bitbuffer  == bitbuffer_t   reference include/bitbuffer.h:34-40   (6604 bytes)
 *   r433_pulse_data == pulse_data_t  reference include/pulse_data.h:30-50  (9672 bytes)
 *   r433_r_device   == r_device      reference include/r_device.h:59-92    (152 bytes)
 *
 * tests/test_abi.py checks the numbers against the reference build (sizeof/offsetof taken from the
 * compiled reference through oracle/_ref).
 */
#ifndef R433_ABI_H_
#define R433_ABI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define R433_BITBUF_COLS 128
#define R433_BITBUF_ROWS 50

typedef struct r433_bitbuffer {
    uint16_t num_rows;                            /* number of active rows */
    uint16_t free_row;                            /* index of next free row */
    uint16_t bits_per_row[R433_BITBUF_ROWS];      /* active bits per row */
    uint16_t syncs_before_row[R433_BITBUF_ROWS];  /* sync pulses seen before each row */
    uint8_t bb[R433_BITBUF_ROWS][R433_BITBUF_COLS];
} r433_bitbuffer;

#define R433_MAX_PULSES 1200

typedef struct r433_pulse_data {
    uint64_t offset;       /* first pulse, in samples from start of stream */
    uint32_t sample_rate;
    unsigned depth_bits;
    unsigned start_ago;
    unsigned end_ago;
    unsigned int num_pulses;
    int pulse[R433_MAX_PULSES];
    int gap[R433_MAX_PULSES];
    int ook_low_estimate;
    int ook_high_estimate;
    int fsk_f1_est;
    int fsk_f2_est;
    float freq1_hz;
    float freq2_hz;
    float centerfreq_hz;
    float range_db;
    float rssi_db;
    float snr_db;
    float noise_db;
} r433_pulse_data;

struct r433_r_device;
struct data; /* the reference's data_t; opaque here */

typedef int (*r433_decode_fn)(struct r433_r_device *decoder, r433_bitbuffer *bitbuffer);

typedef struct r433_r_device {
    unsigned protocol_num;
    char const *name;
    unsigned modulation;  /* enum modulation_types, reference include/r_device.h:24-40 */
    float short_width;    /* us */
    float long_width;     /* us */
    float reset_limit;    /* us */
    float gap_limit;      /* us */
    float sync_width;     /* us */
    float tolerance;      /* us */
    r433_decode_fn decode_fn;
    struct r433_r_device *(*create_fn)(char const *args);
    unsigned priority;    /* run later and only if no earlier level produced an event */
    unsigned disabled;
    char const *const *fields;
    int verbose;
    int verbose_bits;
    void (*log_fn)(struct r433_r_device *decoder, int level, struct data *data);
    void (*output_fn)(struct r433_r_device *decoder, struct data *data);
    unsigned decode_events;
    unsigned decode_ok;
    unsigned decode_messages;
    unsigned decode_fails[5];
    void *decode_ctx;
    void *output_ctx;
} r433_r_device;

/* pulse_detect_fsk_t, reference include/pulse_detect_fsk.h:23-41 (32 bytes) */
typedef struct r433_fsk_state {
    unsigned fsk_pulse_length; /* counter for internal FSK pulse detection */
    unsigned fsk_state;        /* PD_FSK_STATE_INIT 0, _FH 1, _FL 2, _ERROR 3 */
    int fm_f1_est;             /* estimate for the F1 frequency for FSK */
    int fm_f2_est;             /* estimate for the F2 frequency for FSK */
    int16_t var_test_max, var_test_min, maxx, minn, midd; /* min/max detector */
    int skip_samples;
} r433_fsk_state;

/* decode_fn return codes, reference include/r_device.h:45-53 */
#define R433_DECODE_FAIL_OTHER 0
#define R433_DECODE_ABORT_LENGTH (-1)
#define R433_DECODE_ABORT_EARLY (-2)
#define R433_DECODE_FAIL_MIC (-3)
#define R433_DECODE_FAIL_SANITY (-4)

#if defined(__cplusplus)
static_assert(sizeof(r433_bitbuffer) == 6604, "bitbuffer_t layout");
static_assert(sizeof(r433_fsk_state) == 32, "pulse_detect_fsk_t layout");
static_assert(sizeof(r433_pulse_data) == 9672, "pulse_data_t layout");
static_assert(offsetof(r433_pulse_data, pulse) == 28, "pulse_data_t layout");
static_assert(offsetof(r433_pulse_data, ook_low_estimate) == 9628, "pulse_data_t layout");
static_assert(sizeof(void *) != 8 || sizeof(r433_r_device) == 152, "r_device layout");
static_assert(sizeof(void *) != 8 || offsetof(r433_r_device, decode_fn) == 48, "r_device layout");
static_assert(sizeof(void *) != 8 || offsetof(r433_r_device, priority) == 64, "r_device layout");
static_assert(sizeof(void *) != 8 || offsetof(r433_r_device, decode_ctx) == 136, "r_device layout");
#endif

#ifdef __cplusplus
}
#endif
#endif /* R433_ABI_H_ */
