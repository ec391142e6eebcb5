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

// Writes the crate's fixed 96-byte header followed by the payload.
NpyStatus write_npy_payload(const std::string &path, const uint8_t *payload, size_t n, std::string *msg);

}  // namespace msbwt
