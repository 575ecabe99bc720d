// Known-answer tests for the wire formats in front of the engine, with the vectors of the reference's own unit
// tests: android/app/src/test/java/com/vi/slam/android/streaming/ImuDataTest.kt (packet size, field order, little
// endian bytes, round trips, short packet) and FrameMetadataTest.kt (JSON keys, round trips, malformed and
// incomplete JSON, Long.MAX_VALUE sequence numbers, common resolutions).  No GPU is touched: the engine has no
// framework selected, so it drops what it is fed after counting it.
#include "slam/slam_engine.hpp"
#include "streaming/ingest.hpp"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

using namespace vi_slam;

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

static IMUSample imu(int64_t ts, double ax, double ay, double az, double gx, double gy, double gz) {
    IMUSample s;
    s.timestampNs = ts;
    s.acceleration[0] = ax; s.acceleration[1] = ay; s.acceleration[2] = az;
    s.angularVelocity[0] = gx; s.angularVelocity[1] = gy; s.angularVelocity[2] = gz;
    return s;
}
static double le_double(const uint8_t* p) { double d; std::memcpy(&d, p, 8); return d; } /* host is little endian */
static int64_t le_i64(const uint8_t* p) { int64_t v; std::memcpy(&v, p, 8); return v; }

static int test_imu() {
    CHECK(kImuPacketSize == 56); /* ImuDataTest.testPacketSize */
    uint8_t pkt[56];
    /* testToPacket */
    encodeImuPacket(imu(1234567890123456LL, 1.0, 2.0, 3.0, 0.1, 0.2, 0.3), pkt);
    CHECK(le_i64(pkt) == 1234567890123456LL);
    CHECK(le_double(pkt + 8) == 1.0 && le_double(pkt + 16) == 2.0 && le_double(pkt + 24) == 3.0);
    CHECK(le_double(pkt + 32) == 0.1 && le_double(pkt + 40) == 0.2 && le_double(pkt + 48) == 0.3);
    /* testFromPacket */
    {
        const int64_t ts = 9876543210987654LL;
        const double v[6] = {5.0, 6.0, 7.0, 0.5, 0.6, 0.7};
        std::memcpy(pkt, &ts, 8);
        std::memcpy(pkt + 8, v, 48);
        IMUSample s;
        CHECK(decodeImuPacket(pkt, 56, s));
        CHECK(s.timestampNs == ts && s.accX() == 5.0 && s.accY() == 6.0 && s.accZ() == 7.0);
        CHECK(s.gyroX() == 0.5 && s.gyroY() == 0.6 && s.gyroZ() == 0.7);
    }
    /* testRoundTrip, testZeroValues, testNegativeValues: exact, not just within 1e-4 */
    const IMUSample cases[3] = {imu(123456789012345LL, 9.81, 0, 0, 0, 0, 1.57), imu(0, 0, 0, 0, 0, 0, 0),
                                imu(0, -9.81, -1.23, -4.56, -0.1, -0.2, -0.3)};
    for (const IMUSample& c : cases) {
        encodeImuPacket(c, pkt);
        IMUSample d;
        CHECK(decodeImuPacket(pkt, 56, d));
        CHECK(d.timestampNs == c.timestampNs);
        for (int i = 0; i < 3; i++) CHECK(d.acceleration[i] == c.acceleration[i] && d.angularVelocity[i] == c.angularVelocity[i]);
    }
    /* testFromPacketInvalidSize: 50 bytes are rejected (IllegalArgumentException there), output untouched */
    {
        std::vector<uint8_t> small(50, 0);
        IMUSample s = imu(77, 1, 2, 3, 4, 5, 6);
        CHECK(!decodeImuPacket(small.data(), small.size(), s));
        CHECK(s.timestampNs == 77 && s.accX() == 1);
        CHECK(!decodeImuPacket(nullptr, 56, s));
    }
    /* testLittleEndianByteOrder */
    encodeImuPacket(imu(0x0102030405060708LL, 0, 0, 0, 0, 0, 0), pkt);
    CHECK(pkt[0] == 0x08 && pkt[1] == 0x07 && pkt[2] == 0x06 && pkt[3] == 0x05 && pkt[7] == 0x01);
    /* extra bytes behind a packet are ignored (packet.size >= PACKET_SIZE) */
    {
        std::vector<uint8_t> big(64, 0xEE);
        encodeImuPacket(imu(5, 1, 2, 3, 4, 5, 6), big.data());
        IMUSample s;
        CHECK(decodeImuPacket(big.data(), big.size(), s) && s.timestampNs == 5 && s.gyroZ() == 6);
    }
    return 0;
}

static int test_metadata() {
    FrameMetadata m;
    /* testToJson */
    m.sequenceNumber = 12345; m.timestampNs = 1234567890123456LL; m.width = 1920; m.height = 1080; m.exposureTimeNs = 33333333; m.iso = 800;
    const std::string j = encodeFrameMetadata(m);
    for (const char* part : {"\"seq\":12345", "\"ts\":1234567890123456", "\"w\":1920", "\"h\":1080", "\"exp\":33333333", "\"iso\":800"})
        CHECK(j.find(part) != std::string::npos);
    /* testCompactJsonFormat */
    for (const char* verbose : {"sequenceNumber", "timestampNs", "width", "height", "exposureTimeNs"}) CHECK(j.find(verbose) == std::string::npos);
    /* testFromJson */
    FrameMetadata d;
    CHECK(decodeFrameMetadata("{\"seq\":99999,\"ts\":9876543210987654,\"w\":3840,\"h\":2160,\"exp\":16666666,\"iso\":400}", d));
    CHECK(d.sequenceNumber == 99999 && d.timestampNs == 9876543210987654LL && d.width == 3840 && d.height == 2160);
    CHECK(d.exposureTimeNs == 16666666 && d.iso == 400);
    /* testRoundTrip */
    m.sequenceNumber = 54321; m.timestampNs = 998877665544332211LL; m.width = 1280; m.height = 720; m.exposureTimeNs = 20000000; m.iso = 1600;
    CHECK(decodeFrameMetadata(encodeFrameMetadata(m), d));
    CHECK(d.sequenceNumber == m.sequenceNumber && d.timestampNs == m.timestampNs && d.width == 1280 && d.height == 720 &&
          d.exposureTimeNs == 20000000 && d.iso == 1600);
    /* testFromJsonInvalidFormat, testFromJsonMissingFields: rejected, output untouched */
    FrameMetadata keep = d;
    CHECK(!decodeFrameMetadata("{invalid json}", d));
    CHECK(!decodeFrameMetadata("{\"seq\":123}", d));
    CHECK(!decodeFrameMetadata("", d) && !decodeFrameMetadata("[1,2]", d) && !decodeFrameMetadata("{\"seq\":1,\"ts\":2,\"w\":3,\"h\":4} x", d));
    CHECK(!decodeFrameMetadata("{\"seq\":1,\"ts\":2,\"w\":3.5,\"h\":4}", d));
    CHECK(!decodeFrameMetadata("{\"seq\":99999999999999999999,\"ts\":2,\"w\":3,\"h\":4}", d));
    CHECK(d.sequenceNumber == keep.sequenceNumber && d.width == keep.width);
    /* optional fields default to 0 (optLong / optInt); whitespace, key order and unknown keys do not matter */
    CHECK(decodeFrameMetadata(" { \"h\" : 480 , \"w\":640,\n \"note\":\"a,b}\", \"extra\":{\"k\":[1,2]}, \"ts\":1000, \"seq\":0 } ", d));
    CHECK(d.sequenceNumber == 0 && d.timestampNs == 1000 && d.width == 640 && d.height == 480 && d.exposureTimeNs == 0 && d.iso == 0);
    /* testToStringFormat */
    m = FrameMetadata(); m.sequenceNumber = 100; m.timestampNs = 1000000000; m.width = 640; m.height = 480; m.exposureTimeNs = 10000000; m.iso = 200;
    const std::string s = describeFrameMetadata(m);
    for (const char* part : {"seq=100", "ts=1000000000", "640x480", "exp=10000000ns", "iso=200"}) CHECK(s.find(part) != std::string::npos);
    /* testZeroSequenceNumber, testLargeSequenceNumber, testCommonResolutions */
    m = FrameMetadata(); m.timestampNs = 1000; m.width = 1920; m.height = 1080;
    m.sequenceNumber = 0;
    CHECK(decodeFrameMetadata(encodeFrameMetadata(m), d) && d.sequenceNumber == 0);
    m.sequenceNumber = INT64_MAX;
    CHECK(decodeFrameMetadata(encodeFrameMetadata(m), d) && d.sequenceNumber == INT64_MAX);
    const int res[4][2] = {{640, 480}, {1280, 720}, {1920, 1080}, {3840, 2160}};
    for (auto& r : res) {
        m.sequenceNumber = 1; m.width = r[0]; m.height = r[1];
        CHECK(decodeFrameMetadata(encodeFrameMetadata(m), d) && d.width == r[0] && d.height == r[1]);
    }
    return 0;
}

static int test_stream_ingest() {
    SLAMEngine engine; /* no framework selected: processImage / processIMU return before any work */
    StreamIngest in(engine);
    uint8_t dg[3 * 56];
    encodeImuPacket(imu(1000, 0, 0, 9.81, 0, 0, 0), dg);
    encodeImuPacket(imu(2000, 0, 0, 9.81, 0, 0, 0), dg + 56);
    encodeImuPacket(imu(1500, 0, 0, 9.81, 0, 0, 0), dg + 112); /* late duplicate path: older than the last one */
    CHECK(in.onImuDatagram(dg, sizeof(dg)) == 2);
    CHECK(in.counters().imuSamples == 2 && in.counters().imuOutOfOrder == 1);
    CHECK(in.onImuDatagram(dg, 100) == 0 && in.counters().imuBadPackets == 1); /* not a whole number of packets */
    std::vector<uint8_t> px(640 * 480, 128);
    ImageView img(px.data(), 480, 640, 640, 1);
    CHECK(in.onFrame(img, "{\"seq\":10,\"ts\":5000,\"w\":640,\"h\":480}"));
    CHECK(!in.onFrame(img, "{\"seq\":11,\"ts\":6000,\"w\":1920,\"h\":1080}") && in.counters().framesSizeMismatch == 1);
    CHECK(!in.onFrame(img, "{\"seq\":10,\"ts\":7000,\"w\":640,\"h\":480}") && in.counters().framesOutOfOrder == 1);
    CHECK(!in.onFrame(img, "{oops}") && in.counters().framesBadMetadata == 1);
    CHECK(in.onFrame(img, "{\"seq\":14,\"ts\":9000,\"w\":640,\"h\":480,\"exp\":1,\"iso\":2}"));
    CHECK(in.counters().frames == 2 && in.counters().framesLost == 3);
    return 0;
}

int main() {
    if (test_imu() || test_metadata() || test_stream_ingest()) return 1;
    std::printf("all ingest tests passed\n");
    return 0;
}
