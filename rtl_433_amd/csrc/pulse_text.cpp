// pulse_text.cpp -- reader for pulse-data text files (`-r file.ook`), written from the description of the two
// formats such a file may carry, as one pass over the whole text driven by a character-class table.
//
// Format 1, timing lines (what `-w file.ook` writes, src/pulse_data.c:193-224 is the writer):
//     ;<comment>            a header line; "freq1" / "freq2" comments carry the package's two frequencies in Hz
//     <mark> <space>        one pulse and the gap after it, decimal microseconds
//   A header line that follows timing lines closes the package; so does the end of the text.
//
// Format 2, RfRaw lines (the Portisch sniffing firmware's notation, which the analyzer prints as its "RfRaw" hint):
//     AA B0 <len> <n> <repeats> <n bucket widths, 16 bit> <symbols> 55
//     AA B1       <n>           <n bucket widths, 16 bit> <symbols> 55
//   written in hex with optional blanks, dashes and colons between the digits.  A symbol is one hex digit naming a
//   bucket; bit 3 marks a pulse, a digit below 8 is a gap.  Older firmware leaves bit 3 out and alternates pulse, gap
//   by position instead (first digit of a byte = pulse).  Several groups may follow one another on a line, joined by
//   blanks, '+' or '-'; bucket widths are microseconds, so such a package is at 1 MS/s whatever the file name says.
//
// Behaviour is pinned to what the reference CLI decodes from the same files (tests/golden/ook_flex.json,
// tests/test_pulse_door.py); the reader takes untrusted text and never writes past a full package.
#include <array>
#include <climits>

#include "host_common.hpp"

using namespace r433;

namespace {

enum CharClass : uint8_t {
    C_OTHER = 0,
    C_DIGIT = 1,  // 0-9
    C_HEX = 2,    // a-f A-F
    C_BLANK = 4,  // space, tab
    C_JOIN = 8,   // '-' ':' (between hex digits), '+' (between groups)
    C_EOL = 16,   // '\n' '\r'
};

struct ClassTable {
    std::array<uint8_t, 256> cls{};
    std::array<int8_t, 256> val{};
    constexpr ClassTable()
    {
        for (int c = 0; c < 256; ++c)
            val[(size_t)c] = -1;
        for (int c = '0'; c <= '9'; ++c) {
            cls[(size_t)c] = C_DIGIT;
            val[(size_t)c] = (int8_t)(c - '0');
        }
        for (int k = 0; k < 6; ++k) {
            cls[(size_t)('a' + k)] = cls[(size_t)('A' + k)] = C_HEX;
            val[(size_t)('a' + k)] = val[(size_t)('A' + k)] = (int8_t)(10 + k);
        }
        cls[(size_t)' '] = cls[(size_t)'\t'] = C_BLANK;
        cls[(size_t)'-'] = cls[(size_t)':'] = cls[(size_t)'+'] = C_JOIN;
        cls[(size_t)'\n'] = cls[(size_t)'\r'] = C_EOL;
    }
};
constexpr ClassTable kTab;

struct Span {
    char const *at, *end;
    bool empty() const { return at >= end; }
};

// ---- RfRaw ----

// The hex digits of a stretch of text with the filler between them dropped.  `stop` says where the stretch ended in the
// text (a character that is neither a digit nor filler, or the end).
struct Nibbles {
    std::vector<uint8_t> v;
    std::vector<char const *> where; // text position behind each digit
    bool to_the_end = false;         // the last digit was the last character of the stretch
};

void collect_nibbles(Span s, Nibbles &out)
{
    out.v.clear();
    out.where.clear();
    out.to_the_end = false;
    for (char const *p = s.at; p < s.end; ++p) {
        uint8_t const c = (uint8_t)*p;
        if (kTab.val[c] >= 0) {
            out.v.push_back((uint8_t)kTab.val[c]);
            out.where.push_back(p + 1);
            out.to_the_end = p + 1 == s.end;
        }
        else if (!(kTab.cls[c] & (C_BLANK | C_JOIN)) || c == '+') {
            break; // '+' joins groups, it never sits inside one
        }
    }
}

struct Appender {
    r433_pulse_data &d;
    bool open = false; // a pulse is written at d.num_pulses and waits for its gap
    bool full() const { return d.num_pulses >= R433_MAX_PULSES; }
    void mark(int width)
    {
        if (open)
            space(0);
        if (full())
            return;
        d.pulse[d.num_pulses] = width;
        open = true;
    }
    void space(int width)
    {
        if (full())
            return;
        if (!open)
            d.pulse[d.num_pulses] = 0;
        d.gap[d.num_pulses] = width;
        d.num_pulses += 1;
        open = false;
    }
};

// One "AA Bx ... 55" group from the digits at n[from...].  Returns the index behind the group, or 0 when the digits do
// not start a group (nothing is taken then beyond what a broken symbol run already appended).
size_t rfraw_group(Nibbles const &n, size_t from, r433_pulse_data &d, bool &ok)
{
    ok = false;
    size_t i = from;
    auto left = [&]() { return n.v.size() - i; };
    auto byte_at = [&](size_t k) { return (int)(n.v[k] << 4 | n.v[k + 1]); };
    if (left() < 4 || byte_at(i) != 0xaa)
        return 0;
    int const kind = byte_at(i + 2);
    if (kind != 0xb0 && kind != 0xb1)
        return 0;
    i += 4;
    bool const counted = kind == 0xb0; // B0 carries a length byte (not needed: the group ends at its 55) and a repeat count
    size_t const head = counted ? 6 : 2;
    if (left() < head)
        return 0;
    if (counted)
        i += 2;
    int const n_buckets = byte_at(i);
    i += 2;
    int times = 1;
    if (counted) {
        times = byte_at(i);
        i += 2;
    }
    if (n_buckets > 8 || left() < (size_t)n_buckets * 4)
        return 0;
    int bucket[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < n_buckets; ++b, i += 4)
        bucket[b] = byte_at(i) << 8 | byte_at(i + 2);

    // the symbol run: up to a byte-aligned 55 (or the end of the digits)
    size_t run_end = i;
    bool flagged = false; // some symbol byte has a pulse bit: the newer notation
    bool closed = false;
    for (; run_end + 1 < n.v.size(); run_end += 2) {
        int const b = byte_at(run_end);
        if (b == 0x55) {
            closed = true;
            break;
        }
        flagged |= (b & 0x88) != 0;
    }
    size_t const tail = closed ? run_end : n.v.size(); // an odd digit at the very end is still a symbol
    if (d.num_pulses >= R433_MAX_PULSES)
        return 0;
    unsigned const first = d.num_pulses;
    Appender app{d};
    for (size_t k = i; k < tail && !app.full(); ++k) {
        int const sym = n.v[k];
        bool const high_digit = ((k - i) & 1) == 0;
        if (sym >= 8 || (!flagged && high_digit))
            app.mark(bucket[sym & 7]);
        else
            app.space(bucket[sym]);
    }
    // (a pulse still waiting for its gap at the end of the run stays outside the count)
    if (!closed && !n.to_the_end && !app.full())
        return 0; // the run broke off at a foreign character: what it gave stays, but the group does not count as one
    unsigned const made = d.num_pulses - first;
    for (int r = 1; r < times && made > 0 && d.num_pulses + made <= R433_MAX_PULSES; ++r) {
        std::copy(d.pulse + first, d.pulse + first + made, d.pulse + d.num_pulses);
        std::copy(d.gap + first, d.gap + first + made, d.gap + d.num_pulses);
        d.num_pulses += made;
    }
    d.sample_rate = 1000000;
    ok = true;
    return closed ? run_end + 2 : n.v.size();
}

bool starts_rfraw(Span line)
{
    int seen = 0;
    static constexpr uint8_t want[4] = {0xa, 0xa, 0xb, 0x0};
    for (char const *p = line.at; p < line.end && seen < 4; ++p) {
        uint8_t const c = (uint8_t)*p;
        int const v = kTab.val[c];
        if (v >= 0) {
            if ((seen < 3 ? v : (v & ~1)) != want[seen])
                return false;
            seen += 1;
        }
        else if (!(kTab.cls[c] & (C_BLANK | C_JOIN)) || c == '+') {
            return false;
        }
    }
    return seen == 4;
}

void rfraw_line(Span line, r433_pulse_data &d)
{
    Nibbles n;
    char const *p = line.at;
    while (p < line.end && d.num_pulses < R433_MAX_PULSES) {
        // between groups: blanks, line ends, '+' and '-'
        while (p < line.end && ((kTab.cls[(uint8_t)*p] & (C_BLANK | C_EOL)) || *p == '+' || *p == '-'))
            ++p;
        if (p >= line.end)
            break;
        collect_nibbles(Span{p, line.end}, n);
        bool ok = false;
        size_t const used = rfraw_group(n, 0, d, ok);
        if (!ok || used == 0)
            break;
        p = n.where[used - 1];
    }
}

// ---- timing lines ----

// A decimal number the way the C library reads one: blanks, an optional sign, digits.  No digits: zero, and the
// cursor stays where it was.
long take_decimal(Span &s)
{
    char const *p = s.at;
    while (p < s.end && (kTab.cls[(uint8_t)*p] & (C_BLANK | C_EOL) || *p == '\v' || *p == '\f'))
        ++p;
    bool neg = false;
    if (p < s.end && (*p == '+' || *p == '-'))
        neg = *p++ == '-';
    if (p >= s.end || kTab.cls[(uint8_t)*p] != C_DIGIT)
        return 0;
    unsigned long long v = 0;
    for (; p < s.end && kTab.cls[(uint8_t)*p] == C_DIGIT; ++p)
        v = v < (1ull << 62) ? v * 10 + (unsigned)kTab.val[(uint8_t)*p] : v;
    s.at = p;
    long const m = v > (unsigned long long)LONG_MAX ? LONG_MAX : (long)v;
    return neg ? -m : m;
}

bool has_prefix(Span s, char const *word)
{
    size_t const n = strlen(word);
    return (size_t)(s.end - s.at) >= n && !memcmp(s.at, word, n);
}

} // namespace

extern "C" int r433_pulse_text_load(char const *text, size_t len, uint32_t sample_rate, r433_pulse_data *out, uint32_t max_packages)
{
    if ((!text && len) || (!out && max_packages))
        return fail(R433_EINVAL, "null argument");
    double const per_us = sample_rate / 1e6;
    uint32_t n_out = 0;
    r433_pulse_data *pkg = nullptr;
    unsigned filled = 0; // timing lines + RfRaw pairs of the open package
    auto open_package = [&]() -> bool {
        if (n_out >= max_packages)
            return false;
        pkg = &out[n_out];
        memset(pkg, 0, sizeof(*pkg));
        pkg->sample_rate = sample_rate;
        filled = 0;
        return true;
    };
    auto close_package = [&]() {
        pkg->num_pulses = filled;
        n_out += 1;
        pkg = nullptr;
    };
    char const *p = text, *const end = text + len;
    bool stop = !open_package();
    while (!stop && p < end) {
        char const *eol = (char const *)memchr(p, '\n', (size_t)(end - p));
        Span line{p, eol ? eol + 1 : end}; // the line with its newline
        p = line.end;
        if (*line.at == ';') {
            Span rest{line.at + 6, line.end};
            if (has_prefix(line, ";freq1"))
                pkg->freq1_hz = (float)take_decimal(rest);
            else if (has_prefix(line, ";freq2"))
                pkg->freq2_hz = (float)take_decimal(rest);
            if (filled) { // a header after data: this package is complete, the line belongs to nobody
                close_package();
                stop = !open_package();
            }
            continue;
        }
        if (starts_rfraw(line)) {
            // (the reference keeps its count of timing lines apart from the package's own counter, which only RfRaw lines
            // move: such a line after timing lines starts over at the package's first slot.  Nobody writes such files.)
            rfraw_line(line, *pkg);
            filled = pkg->num_pulses;
        }
        else {
            Span cur = line;
            long const mark = take_decimal(cur);
            cur.at = cur.at < cur.end ? cur.at + 1 : cur.end; // one separator character
            long const space = take_decimal(cur);
            if (mark >= 0 && space >= 0) { // a negative width is a damaged line: left out
                pkg->pulse[filled] = (int)(per_us * mark);
                pkg->gap[filled] = (int)(per_us * space);
                filled += 1;
            }
        }
        if (filled >= R433_MAX_PULSES) { // a full package is complete as it is
            close_package();
            stop = !open_package();
        }
    }
    if (pkg && filled)
        close_package();
    return (int)n_out;
}
