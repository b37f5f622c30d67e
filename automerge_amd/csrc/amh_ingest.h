#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

bool amh_ingest_normalise(const uint8_t* p, size_t len, std::vector<uint8_t>& out);
void amh_host_sha256(const uint8_t* a, size_t an, const uint8_t* b, size_t bn, uint8_t* out32);
bool amh_host_deflate_document(const uint8_t* p, size_t len, std::vector<uint8_t>& out);
bool amh_ingest_inflate_change(const uint8_t* p, size_t len, std::vector<uint8_t>& out);
