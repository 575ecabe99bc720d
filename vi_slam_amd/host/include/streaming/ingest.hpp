// Wire formats in front of the engine, as the phone app of the reference produces them:
//   * IMU sample: 56-byte little-endian packet -- int64 timestamp_ns, double acc[3] (m/s^2), double gyro[3] (rad/s)
//     (android/app/src/main/java/com/vi/slam/android/streaming/ImuData.kt:38-90);
//   * frame metadata: compact JSON object {"seq","ts","w","h","exp","iso"} sent next to the video stream
//     (android/app/src/main/java/com/vi/slam/android/streaming/FrameMetadata.kt:24-85).
// Decoders never throw; they return false and leave the output untouched on malformed input (the Kotlin side
// throws IllegalArgumentException / JSONException in the same cases).
#pragma once
#include "common/types.hpp"
#include "slam/image_view.hpp"
#include <cstddef>
#include <cstdint>
#include <string>

namespace vi_slam {

class SLAMEngine;

constexpr std::size_t kImuPacketSize = 56;

/* ImuData.fromPacket: needs at least 56 bytes; extra bytes are ignored */
bool decodeImuPacket(const uint8_t* packet, std::size_t size, IMUSample& out);
/* ImuData.toPacket: writes exactly 56 bytes */
void encodeImuPacket(const IMUSample& imu, uint8_t* packet56);

struct FrameMetadata {
    int64_t sequenceNumber = 0;
    int64_t timestampNs = 0;
    int width = 0, height = 0;
    int64_t exposureTimeNs = 0; /* optional in the JSON, default 0 */
    int iso = 0;                /* optional in the JSON, default 0 */
};

/* FrameMetadata.fromJson: "seq", "ts", "w", "h" are required integers, "exp" and "iso" optional; unknown keys are
 * skipped; anything that is not one flat JSON object fails */
bool decodeFrameMetadata(const std::string& json, FrameMetadata& out);
/* FrameMetadata.toJson: {"seq":..,"ts":..,"w":..,"h":..,"exp":..,"iso":..} without whitespace */
std::string encodeFrameMetadata(const FrameMetadata& m);
/* FrameMetadata.toString */
std::string describeFrameMetadata(const FrameMetadata& m);

/* Feeds decoded packets into an engine: the glue the reference's PC client leaves as a TODO in its video
 * callback (pc_client/src/main.cpp:182-192).  Counts what it dropped and why. */
class StreamIngest {
public:
    explicit StreamIngest(SLAMEngine& engine) : engine_(engine) {}
    /* one UDP datagram = one or more 56-byte IMU packets back to back; returns the number of samples fed */
    int onImuDatagram(const uint8_t* data, std::size_t size);
    /* a decoded video frame plus its metadata JSON: the frame is processed at the metadata's sensor timestamp if
     * the sizes agree and the sequence number moved forward */
    bool onFrame(const ImageView& image, const std::string& metadataJson);
    struct Counters {
        int64_t imuSamples = 0, imuBadPackets = 0, imuOutOfOrder = 0;
        int64_t frames = 0, framesBadMetadata = 0, framesSizeMismatch = 0, framesOutOfOrder = 0, framesLost = 0;
    };
    const Counters& counters() const { return c_; }

private:
    SLAMEngine& engine_;
    Counters c_;
    int64_t lastImuNs_ = INT64_MIN, lastSeq_ = -1;
};

}  // namespace vi_slam
