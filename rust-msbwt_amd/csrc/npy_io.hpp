// NumPy v1.0 container used for comp_msbwt.npy (load path; mirrors what
// src/rle_bwt.rs:81-155 accepts/rejects and what src/bwt_converter.rs:102-184 writes).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace msbwt {

enum class NpyStatus { kOk, kIo, kUnexpectedEof, kBadHeader };

// Reads the u1 payload of `path` into *payload.  *msg gets a human-readable reason.
NpyStatus read_npy_payload(const std::string &path, std::vector<uint8_t> *payload, std::string *msg);

// Read-only memory map of a file's payload (no host copy of a multi-GB comp_msbwt.npy: the
// bytes go from the page cache straight to the device upload).
class MappedPayload {
  public:
    MappedPayload() = default;
    ~MappedPayload();
    MappedPayload(const MappedPayload &) = delete;
    MappedPayload &operator=(const MappedPayload &) = delete;
    const uint8_t *data() const { return payload_; }
    size_t size() const { return size_; }

  private:
    friend NpyStatus map_npy_payload(const std::string &path, MappedPayload *out, std::string *msg);
    void *base_ = nullptr;
    size_t map_len_ = 0;
    const uint8_t *payload_ = nullptr;
    size_t size_ = 0;
};

// Same checks as read_npy_payload, but maps the payload instead of reading it.
NpyStatus map_npy_payload(const std::string &path, MappedPayload *out, std::string *msg);

// Writes the crate's fixed 96-byte header followed by the payload.
NpyStatus write_npy_payload(const std::string &path, const uint8_t *payload, size_t n, std::string *msg);

}  // namespace msbwt
