// containers.cpp — snarkjs binary containers (.zkey / .wtns) and the prover host's error text.
//   FileWrapper::read_bin_file ← src/file_wrapper.rs:45-103;  read_wtns_header ← :169-177, src/proof_helper.rs:247-268
#include <algorithm>
#include <fcntl.h>
#include <errno.h>
#include <sys/mman.h>
#include <sys/random.h>
#include <sys/stat.h>
#include <unistd.h>

#include "prover_internal.h"

using namespace bn254;
using namespace isnark;
using namespace isnark::prover;

namespace isnark {
namespace prover {

static thread_local char g_perr[512] = "";
int fail(int code, const char* fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_perr, sizeof g_perr, fmt, ap);
  va_end(ap);
  return code;
}
const char* last_error_text() { return g_perr; }
void set_error_text(const char* text) { snprintf(g_perr, sizeof g_perr, "%s", text ? text : ""); }

// FileWrapper::read_bin_file — src/file_wrapper.rs:45-103
int read_sections(const uint8_t* data, size_t len, const char* type, uint32_t max_version, std::vector<Section>& out)
{
  if (len < 12 || memcmp(data, type, 4) != 0) return fail(ERR_FORMAT, "Invalid File format (expected '%s')", type);
  uint32_t version, nsec;
  memcpy(&version, data + 4, 4);
  memcpy(&nsec, data + 8, 4);
  if (version > max_version) return fail(ERR_FORMAT, "Version not supported");
  out.assign(nsec + 1 > 16 ? nsec + 1 : 16, Section());
  size_t pos = 12;
  for (uint32_t i = 0; i < nsec; i++) {
    if (len - pos < 12) return fail(ERR_FORMAT, "truncated section table");
    uint32_t ht;
    uint64_t hl;
    memcpy(&ht, data + pos, 4);
    memcpy(&hl, data + pos + 4, 8);
    pos += 12;
    if (hl > len - pos) return fail(ERR_FORMAT, "section %u exceeds the file", ht); // pos <= len here; `pos + hl` could wrap for a hostile 64-bit length
    if (ht < out.size()) {
      out[ht].p = data + pos;
      out[ht].size = hl;
      out[ht].count++;
    }
    pos += hl;
  }
  return 0;
}
int unique_section(const std::vector<Section>& s, size_t id, const Section** sec)
{
  if (id >= s.size() || s[id].count == 0) return fail(ERR_FORMAT, "Missing section %zu", id);
  if (s[id].count > 1) return fail(ERR_FORMAT, "Section Duplicated %zu", id);
  *sec = &s[id];
  return 0;
}

MappedFile::~MappedFile()
{
  if (data) munmap((void*)data, len);
  if (fd >= 0) close(fd);
}
int MappedFile::open_ro(const char* path)
  {
    fd = ::open(path, O_RDONLY); // the reference opens read-write although it only reads (file_wrapper.rs:50-54)
    if (fd < 0) return fail(ERR_IO, "cannot open %s", path);
    struct stat st;
    if (fstat(fd, &st) != 0) return fail(ERR_IO, "cannot stat %s", path);
    len = (size_t)st.st_size;
    void* p = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
    if (p == MAP_FAILED) return fail(ERR_IO, "cannot mmap %s", path);
    data = (const uint8_t*)p;
    (void)madvise(p, len, MADV_WILLNEED); // start the read-ahead; the upload workers touch the pages in parallel
    return 0;
  }

// read_wtns_header + section 2 — src/file_wrapper.rs:169-177, src/proof_helper.rs:247-268
int parse_wtns(const uint8_t* data, size_t len, Wtns& w)
{
  std::vector<Section> s;
  if (int rc = read_sections(data, len, "wtns", 2, s)) return rc;
  const Section *h, *v;
  if (int rc = unique_section(s, 1, &h)) return rc;
  if (int rc = unique_section(s, 2, &v)) return rc;
  if (h->size < 8) return fail(ERR_FORMAT, "wtns header too short");
  memcpy(&w.n8, h->p, 4);
  if (w.n8 != 32 || h->size != 4 + 32 + 4) return fail(ERR_FORMAT, "wtns: unsupported field size %u", w.n8);
  memcpy(w.q.l, h->p + 4, 32);
  memcpy(&w.n_witness, h->p + 36, 4);
  if (v->size != (uint64_t)w.n_witness * 32) return fail(ERR_FORMAT, "wtns: section 2 size mismatch");
  w.values = v->p;
  return 0;
}

} // namespace prover
} // namespace isnark
