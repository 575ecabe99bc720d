#include "streaming/ingest.hpp"
#include "slam/slam_engine.hpp"
#include <cstring>

namespace vi_slam {

namespace {
uint64_t loadLe64(const uint8_t* p) {
    uint64_t v = 0;
    for (int i = 7; i >= 0; i--) v = (v << 8) | p[i];
    return v;
}
void storeLe64(uint64_t v, uint8_t* p) {
    for (int i = 0; i < 8; i++) { p[i] = (uint8_t)(v & 0xFF); v >>= 8; }
}
double bitsToDouble(uint64_t b) { double d; std::memcpy(&d, &b, 8); return d; }
uint64_t doubleToBits(double d) { uint64_t b; std::memcpy(&b, &d, 8); return b; }

/* minimal reader for one flat JSON object with string keys and integer values (other value kinds are skipped) */
struct JsonCursor {
    const char* p; const char* end;
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
    bool eat(char c) { ws(); if (p < end && *p == c) { p++; return true; } return false; }
    bool string(std::string& out) {
        ws();
        if (p >= end || *p != '"') return false;
        p++;
        out.clear();
        while (p < end && *p != '"') {
            if (*p == '\\') { p++; if (p >= end) return false; }
            out.push_back(*p++);
        }
        if (p >= end) return false;
        p++;
        return true;
    }
    /* integer (optionally with a fraction/exponent that is ignored after the integer part) */
    bool integer(int64_t& out) {
        ws();
        const char* s = p;
        bool neg = false;
        if (p < end && (*p == '-' || *p == '+')) { neg = *p == '-'; p++; }
        if (p >= end || *p < '0' || *p > '9') { p = s; return false; }
        uint64_t v = 0;
        while (p < end && *p >= '0' && *p <= '9') {
            const uint64_t d = (uint64_t)(*p - '0');
            if (v > (UINT64_MAX - d) / 10) return false;
            v = v * 10 + d; p++;
        }
        if (p < end && (*p == '.' || *p == 'e' || *p == 'E')) return false;
        if (!neg && v > (uint64_t)INT64_MAX) return false;
        if (neg && v > (uint64_t)INT64_MAX + 1) return false;
        out = neg ? (int64_t)(0 - v) : (int64_t)v;
        return true;
    }
    /* skips a string, number, literal, or a nested object/array */
    bool skipValue() {
        ws();
        if (p >= end) return false;
        if (*p == '"') { std::string s; return string(s); }
        if (*p == '{' || *p == '[') {
            int depth = 0;
            bool inStr = false;
            for (; p < end; p++) {
                if (inStr) { if (*p == '\\') p++; else if (*p == '"') inStr = false; continue; }
                if (*p == '"') inStr = true;
                else if (*p == '{' || *p == '[') depth++;
                else if (*p == '}' || *p == ']') { if (--depth == 0) { p++; return true; } }
            }
            return false;
        }
        const char* s = p;
        while (p < end && *p != ',' && *p != '}' && *p != ' ' && *p != '\n' && *p != '\r' && *p != '\t') p++;
        return p > s;
    }
};
}  // namespace

bool decodeImuPacket(const uint8_t* packet, std::size_t size, IMUSample& out) {
    if (!packet || size < kImuPacketSize) return false;
    IMUSample s;
    s.timestampNs = (int64_t)loadLe64(packet);
    for (int i = 0; i < 3; i++) s.acceleration[i] = bitsToDouble(loadLe64(packet + 8 + 8 * i));
    for (int i = 0; i < 3; i++) s.angularVelocity[i] = bitsToDouble(loadLe64(packet + 32 + 8 * i));
    out = s;
    return true;
}

void encodeImuPacket(const IMUSample& imu, uint8_t* p) {
    storeLe64((uint64_t)imu.timestampNs, p);
    for (int i = 0; i < 3; i++) storeLe64(doubleToBits(imu.acceleration[i]), p + 8 + 8 * i);
    for (int i = 0; i < 3; i++) storeLe64(doubleToBits(imu.angularVelocity[i]), p + 32 + 8 * i);
}

bool decodeFrameMetadata(const std::string& json, FrameMetadata& out) {
    JsonCursor c{json.data(), json.data() + json.size()};
    if (!c.eat('{')) return false;
    FrameMetadata m;
    bool hasSeq = false, hasTs = false, hasW = false, hasH = false;
    if (!c.eat('}')) {
        for (;;) {
            std::string key;
            if (!c.string(key) || !c.eat(':')) return false;
            int64_t v = 0;
            const bool known = key == "seq" || key == "ts" || key == "w" || key == "h" || key == "exp" || key == "iso";
            if (known) {
                if (!c.integer(v)) return false;
                if (key == "seq") { m.sequenceNumber = v; hasSeq = true; }
                else if (key == "ts") { m.timestampNs = v; hasTs = true; }
                else if (key == "exp") m.exposureTimeNs = v;
                else {
                    if (v < INT32_MIN || v > INT32_MAX) return false;
                    if (key == "w") { m.width = (int)v; hasW = true; }
                    else if (key == "h") { m.height = (int)v; hasH = true; }
                    else m.iso = (int)v;
                }
            } else if (!c.skipValue()) return false;
            if (c.eat(',')) continue;
            if (c.eat('}')) break;
            return false;
        }
    }
    c.ws();
    if (c.p != c.end) return false;
    if (!(hasSeq && hasTs && hasW && hasH)) return false;
    out = m;
    return true;
}

std::string encodeFrameMetadata(const FrameMetadata& m) {
    return "{\"seq\":" + std::to_string(m.sequenceNumber) + ",\"ts\":" + std::to_string(m.timestampNs) + ",\"w\":" +
           std::to_string(m.width) + ",\"h\":" + std::to_string(m.height) + ",\"exp\":" + std::to_string(m.exposureTimeNs) +
           ",\"iso\":" + std::to_string(m.iso) + "}";
}

std::string describeFrameMetadata(const FrameMetadata& m) {
    return "FrameMetadata(seq=" + std::to_string(m.sequenceNumber) + ", ts=" + std::to_string(m.timestampNs) + ", " +
           std::to_string(m.width) + "x" + std::to_string(m.height) + ", exp=" + std::to_string(m.exposureTimeNs) + "ns, iso=" +
           std::to_string(m.iso) + ")";
}

int StreamIngest::onImuDatagram(const uint8_t* data, std::size_t size) {
    if (!data || size < kImuPacketSize || size % kImuPacketSize != 0) { c_.imuBadPackets++; return 0; }
    int fed = 0;
    for (std::size_t o = 0; o + kImuPacketSize <= size; o += kImuPacketSize) {
        IMUSample s;
        if (!decodeImuPacket(data + o, kImuPacketSize, s)) { c_.imuBadPackets++; continue; }
        if (s.timestampNs <= lastImuNs_) { c_.imuOutOfOrder++; continue; } /* UDP may reorder or duplicate */
        lastImuNs_ = s.timestampNs;
        engine_.processIMU(s);
        c_.imuSamples++; fed++;
    }
    return fed;
}

bool StreamIngest::onFrame(const ImageView& image, const std::string& metadataJson) {
    FrameMetadata m;
    if (!decodeFrameMetadata(metadataJson, m)) { c_.framesBadMetadata++; return false; }
    if (m.width != image.cols || m.height != image.rows) { c_.framesSizeMismatch++; return false; }
    if (m.sequenceNumber <= lastSeq_) { c_.framesOutOfOrder++; return false; }
    if (lastSeq_ >= 0 && m.sequenceNumber > lastSeq_ + 1) c_.framesLost += m.sequenceNumber - lastSeq_ - 1;
    lastSeq_ = m.sequenceNumber;
    engine_.processImage(image, m.timestampNs);
    c_.frames++;
    return true;
}

}  // namespace vi_slam
