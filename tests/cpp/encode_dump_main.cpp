// Test driver (CPU only): encode_dump <file.tbl> <chunk_size> <unencoded|dictionary|for> prints, per chunk and column, the
// buffers the host mirror's load_table + ChunkEncoder produce:
//   <chunk> <column> <kind> <size> <width> <data hex> <aux hex or entries> <null words hex or ->
// tests/test_host_encoders.py compares them byte for byte with hyrise_amd/storage.py's encoders (which tests/golden's Hyrise
// .bin exports pin).
#include <cstdio>
#include <cstring>
#include <string>

#include "../../hyrise_amd/host/hyrise_host.hpp"

using namespace hyrise_amd;

static void hex(const void* data, size_t bytes) {
  if (bytes == 0) { std::printf("-"); return; }
  const auto* p = static_cast<const unsigned char*>(data);
  for (size_t i = 0; i < bytes; ++i) std::printf("%02x", p[i]);
}

template <typename T> static void dictionary_entries(const std::vector<T>& d) { hex(d.data(), d.size() * sizeof(T)); }
template <> void dictionary_entries<std::string>(const std::vector<std::string>& d) {
  if (d.empty()) { std::printf("-"); return; }
  for (size_t i = 0; i < d.size(); ++i) { if (i) std::printf(","); if (d[i].empty()) std::printf("."); else hex(d[i].data(), d[i].size()); }
}

template <typename T> static bool dump_typed(const AbstractSegment* segment) {
  if (const auto* value = dynamic_cast<const ValueSegment<T>*>(segment)) {
    std::printf("value %u %zu ", value->size(), sizeof(T));
    if constexpr (std::is_same_v<T, std::string>) { dictionary_entries(value->values()); std::printf(" - "); }
    else { hex(value->values().data(), value->values().size() * sizeof(T)); std::printf(" - "); }
    if (value->is_nullable()) hex(value->null_words().data(), value->null_words().size() * 8); else std::printf("-");
    return true;
  }
  if (const auto* dict = dynamic_cast<const DictionarySegment<T>*>(segment)) {
    std::printf("dictionary %u %u ", dict->size(), dict->attribute_vector().width);
    hex(dict->attribute_vector().bytes.data(), size_t{dict->size()} * dict->attribute_vector().width);
    std::printf(" ");
    dictionary_entries(dict->dictionary());
    std::printf(" -");
    return true;
  }
  return false;
}

int main(int argc, char** argv) {
  if (argc != 4) return 2;
  const auto table = load_table(argv[1], static_cast<ChunkOffset>(std::stoul(argv[2])));
  const std::string encoding = argv[3];
  ChunkEncoder::encode_all_chunks(table, encoding == "dictionary" ? EncodingType::Dictionary
                                         : encoding == "for"      ? EncodingType::FrameOfReference
                                                                  : EncodingType::Unencoded);
  for (ChunkID chunk_id = 0; chunk_id < table->chunk_count(); ++chunk_id) {
    for (ColumnID column = 0; column < table->column_count(); ++column) {
      const auto segment = table->get_chunk(chunk_id)->get_segment(column);
      std::printf("%u %u ", chunk_id, static_cast<unsigned>(column));
      if (const auto* frame = dynamic_cast<const FrameOfReferenceSegment*>(segment.get())) {
        std::printf("for %u %u ", frame->size(), frame->offset_values().width);
        hex(frame->offset_values().bytes.data(), size_t{frame->size()} * frame->offset_values().width);
        std::printf(" ");
        hex(frame->block_minima().data(), frame->block_minima().size() * 4);
        std::printf(" ");
        if (frame->has_nulls()) hex(frame->null_words().data(), frame->null_words().size() * 8); else std::printf("-");
      } else if (!dump_typed<int32_t>(segment.get()) && !dump_typed<int64_t>(segment.get()) && !dump_typed<float>(segment.get()) &&
                 !dump_typed<double>(segment.get()) && !dump_typed<std::string>(segment.get())) {
        return 3;
      }
      std::printf("\n");
    }
  }
  return 0;
}
