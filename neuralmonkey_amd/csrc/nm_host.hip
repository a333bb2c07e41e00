// libnmhip: host-side utilities of the C ABI (no device code).
//
//   nm_crc32c   CRC-32C (Castagnoli) of a host buffer -- the checksum TensorFlow's tensor-bundle
//               checkpoints carry per tensor and per index block (checkpoint import / export,
//               neuralmonkey/tf_manager.py:274-288 -> tf.train.Saver).  Slicing-by-8, ~1 GB/s.
#include "nm_common.h"

static uint32_t g_crc_table[8][256];
static bool g_crc_ready = false;

static void crc32c_init() {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
        g_crc_table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int t = 1; t < 8; ++t)
            g_crc_table[t][i] = (g_crc_table[t - 1][i] >> 8) ^ g_crc_table[0][g_crc_table[t - 1][i] & 0xff];
    g_crc_ready = true;
}

// crc = nm_crc32c(crc_of_previous_bytes, data, n); start with 0
extern "C" uint32_t nm_crc32c(uint32_t crc, const void* data, int64_t n) {
    if (!g_crc_ready) crc32c_init();
    const uint8_t* p = static_cast<const uint8_t*>(data);
    uint32_t c = crc ^ 0xFFFFFFFFu;
    while (n > 0 && (reinterpret_cast<uintptr_t>(p) & 7)) {
        c = g_crc_table[0][(c ^ *p++) & 0xff] ^ (c >> 8);
        --n;
    }
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        w ^= c;
        c = g_crc_table[7][w & 0xff] ^ g_crc_table[6][(w >> 8) & 0xff] ^ g_crc_table[5][(w >> 16) & 0xff] ^
            g_crc_table[4][(w >> 24) & 0xff] ^ g_crc_table[3][(w >> 32) & 0xff] ^ g_crc_table[2][(w >> 40) & 0xff] ^
            g_crc_table[1][(w >> 48) & 0xff] ^ g_crc_table[0][(w >> 56) & 0xff];
        p += 8;
        n -= 8;
    }
    while (n-- > 0) c = g_crc_table[0][(c ^ *p++) & 0xff] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
