// Native Kaldi ark batch reader / vector-ark formatter (host side of the extraction path).
//
// The reference parses one record at a time in Python (dataset/kaldi_io.py: read_key :694-707 reads the
// key one byte per call, _read_mat_binary :1014-1031, _read_compressed_mat :1071-1115, write_vec_flt
// :915-946).  At >100 k utterances/s per GPU that per-record interpreter work is the bottleneck, so this
// file parses a whole batch of matrix records straight into ONE caller-provided buffer (pinned host
// memory on the product path) and formats a whole batch of output vectors in one call.  Same wire format:
//   key SP \0 B  'FM '|'DM '  \4 <i32 rows> \4 <i32 cols> payload        (float / double matrix)
//   key SP \0 B  'CM '  <f32 min><f32 range><i32 rows><i32 cols>  cols x 4 x u16, col-major u8 payload
//   key SP \0 B  'FV '  \4 <i32 dim> payload                             (output vectors)
// Plain C ABI (include/xvec_hip.h), no HIP calls: usable and tested without a GPU.
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>
#include <thread>
#include <vector>

#include <algorithm>
#include <atomic>
#include <mutex>

#include "../../include/xvec_hip.h"

struct xv_ark_reader {
  int fd = -1;
  bool own_fd = false;
  std::vector<unsigned char> buf;   // read-ahead window (streams: pipes, descriptors)
  const unsigned char* win = nullptr;   // base of the bytes [pos, end): buf.data(), or the mapping of a regular file
  size_t pos = 0, end = 0;
  // a regular file opened by name is mapped: headers are parsed in place (no read() per record, no window to refill) and
  // the float payloads of a batch go from the page cache into the caller's (pinned) buffer by pread() from a few
  // threads -- one copy, no page faults on the payload pages of the mapping.  (Like every mmap reader: an ark that is
  // TRUNCATED by another process while a job reads it ends that job with SIGBUS on the next header access, where the
  // read() path reports "truncated"; payloads go through pread and report an error.  Arks are written once by the
  // feature stage before extraction starts -- run_extract_embeddings.sh:43-47 -- so this is not a case the path meets.)
  void* map = nullptr;
  size_t map_len = 0;
  // mapped float payloads of the batch being assembled: copied at the end of xv_ark_next_batch, by a few threads
  struct Copy { int64_t off; float* dst; size_t bytes; };
  std::atomic<bool> copy_failed{false};   // set by any of the copy threads of run_copies
  std::vector<Copy> copies;
  int copy_threads = 1;
  bool eof = false;
  std::string err;
  // one parsed-but-not-yet-delivered record header (when it did not fit the caller's batch)
  bool have_pending = false;
  std::string pending_key;
  int pending_kind = 0;             // 1 FM, 2 DM, 3 CM
  int32_t pending_rows = 0, pending_cols = 0;
  float pending_min = 0.f, pending_range = 0.f;
  int64_t skipped_short = 0;
  int64_t file_pos = 0;             // file offset of buf[end] (bytes consumed from fd so far)
  // scp mode (xv_ark_open_scp): the table of `key rxfile` lines; records are reached by seeking
  struct ScpEntry { std::string key; int path; int64_t offset; };
  bool scp = false;
  std::vector<ScpEntry> entries;
  std::vector<std::string> paths;
  size_t next_entry = 0;
  int cur_path = -1;
};

namespace {

constexpr size_t kChunk = 64u << 10;     // read-ahead window: headers only; float payloads bypass it

// make sure n bytes are available at r->pos (false at EOF / error)
bool fill(xv_ark_reader* r, size_t n) {
  if (r->end - r->pos >= n) return true;
  if (r->map) return false;              // mapped file: [pos, end) is all there is
  if (r->pos > 0) {
    memmove(r->buf.data(), r->win + r->pos, r->end - r->pos);
    r->end -= r->pos;
    r->pos = 0;
  }
  if (r->buf.size() < n) r->buf.resize(n + kChunk);
  while (r->end < n && !r->eof) {
    const ssize_t got = read(r->fd, r->buf.data() + r->end, r->buf.size() - r->end);
    if (got < 0) {
      if (errno == EINTR) continue;
      r->err = std::string("read failed: ") + strerror(errno);
      r->eof = true;
      break;
    }
    if (got == 0) { r->eof = true; break; }
    r->end += (size_t)got;
    r->file_pos += got;
  }
  r->win = r->buf.data();
  return r->end - r->pos >= n;
}

// threads for the payload copies of a mapped ark (a job of the launcher is one process per GPU; 16 host cores per GPU)
int default_copy_threads() {
  const unsigned hw = std::thread::hardware_concurrency();
  return hw >= 8 ? 4 : (hw >= 4 ? 2 : 1);
}

void unmap_file(xv_ark_reader* r) {
  if (r->map) munmap(r->map, r->map_len);
  r->map = nullptr;
  r->map_len = 0;
  r->win = r->buf.data();
}

// map r->fd if it is a non-empty regular file (else the read() path stays in use); `sequential` = whole-ark scan
void map_file(xv_ark_reader* r, bool sequential) {
  unmap_file(r);
  struct stat st;
  if (fstat(r->fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0) return;
  void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, r->fd, 0);
  if (m == MAP_FAILED) return;
  (void)sequential;
  r->map = m;
  r->map_len = (size_t)st.st_size;
  r->win = static_cast<const unsigned char*>(m);
  r->pos = 0;
  r->end = r->map_len;
  r->file_pos = (int64_t)r->map_len;
  r->eof = true;
}

// run the noted payload copies: contiguous shares of the byte total per thread (a share may start inside a record)
void run_copies(xv_ark_reader* r) {
  std::vector<xv_ark_reader::Copy>& c = r->copies;
  if (c.empty()) return;
  size_t total = 0;
  for (const auto& x : c) total += x.bytes;
  const int nt = total >= ((size_t)2 << 20) ? r->copy_threads : 1;
  const int fd = r->fd;
  std::atomic<bool>* failed = &r->copy_failed;
  auto work = [&c, fd, failed](size_t lo, size_t hi) {     // bytes [lo, hi) of the concatenation of all copies
    size_t at = 0;
    for (const auto& x : c) {
      if (hi <= at) break;
      size_t a = lo > at ? lo - at : 0;
      const size_t b = (hi - at < x.bytes) ? hi - at : x.bytes;
      while (a < b) {
        const ssize_t got = pread(fd, reinterpret_cast<unsigned char*>(x.dst) + a, b - a, (off_t)(x.off + (int64_t)a));
        if (got < 0 && errno == EINTR) continue;
        if (got <= 0) { failed->store(true, std::memory_order_relaxed); return; }
        a += (size_t)got;
      }
      at += x.bytes;
    }
  };
  if (nt <= 1) {
    work(0, total);
  } else {
    std::vector<std::thread> th;
    const size_t share = ((total + nt - 1) / nt + 4095) & ~(size_t)4095;
    for (int t = 1; t < nt; ++t) {
      const size_t lo = share * t, hi = lo + share < total ? lo + share : total;
      if (lo < hi) th.emplace_back(work, lo, hi);
    }
    work(0, share < total ? share : total);
    for (auto& x : th) x.join();
  }
  c.clear();
}

int fail(xv_ark_reader* r, const char* msg) {
  r->err = msg;
  return XV_ERR_INVALID;
}

// record header after the key ("\0B" + type + dimensions) -> pending_*; 1 = parsed, <0 error.
// Float vectors ('FV ' / 'DV ', e.g. per-frame VAD decisions) are delivered as [dim, 1] matrices.
int parse_matrix_header(xv_ark_reader* r, std::string& key) {
  if (!fill(r, 5)) return fail(r, "unexpected end of stream after the key");
  const unsigned char* p = r->win + r->pos;
  if (p[0] != 0 || p[1] != 'B') return fail(r, "text-mode or unknown record (expected \\\\0B)");
  int kind = 0;
  bool vec = false;
  if (!memcmp(p + 2, "FM ", 3)) kind = 1;
  else if (!memcmp(p + 2, "DM ", 3)) kind = 2;
  else if (!memcmp(p + 2, "CM ", 3)) kind = 3;
  else if (!memcmp(p + 2, "FV ", 3)) { kind = 1; vec = true; }
  else if (!memcmp(p + 2, "DV ", 3)) { kind = 2; vec = true; }
  else return fail(r, "unknown matrix header (FM / DM / CM / FV / DV are supported)");
  r->pos += 5;
  if (kind == 3) {
    if (!fill(r, 16)) return fail(r, "truncated compressed-matrix header");
    const unsigned char* q = r->win + r->pos;
    memcpy(&r->pending_min, q, 4);
    memcpy(&r->pending_range, q + 4, 4);
    memcpy(&r->pending_rows, q + 8, 4);
    memcpy(&r->pending_cols, q + 12, 4);
    r->pos += 16;
  } else if (vec) {
    if (!fill(r, 5)) return fail(r, "truncated vector header");
    const unsigned char* q = r->win + r->pos;
    if (q[0] != 4) return fail(r, "vector header: int-size marker missing");
    memcpy(&r->pending_rows, q + 1, 4);
    r->pending_cols = 1;
    r->pos += 5;
  } else {
    if (!fill(r, 10)) return fail(r, "truncated matrix header");
    const unsigned char* q = r->win + r->pos;
    if (q[0] != 4 || q[5] != 4) return fail(r, "matrix header: int-size markers missing");
    memcpy(&r->pending_rows, q + 1, 4);
    memcpy(&r->pending_cols, q + 6, 4);
    r->pos += 10;
  }
  if (r->pending_rows < 0 || r->pending_cols < 0) return fail(r, "negative matrix dimension");
  r->pending_key.swap(key);
  r->pending_kind = kind;
  r->have_pending = true;
  return 1;
}

// next record header of a sequential ark -> pending_*; returns 1 = header parsed, 0 = clean end of stream, <0 error
int parse_header(xv_ark_reader* r) {
  // key: bytes up to the first space
  std::string key;
  for (;;) {
    if (!fill(r, 1)) {
      if (!r->err.empty()) return XV_ERR_INVALID;
      bool blank = true;
      for (char c : key) if (c != '\n' && c != ' ' && c != '\t' && c != '\r') blank = false;
      if (blank) return 0;
      return fail(r, "unexpected end of stream inside a key");
    }
    const unsigned char* p = r->win + r->pos;
    const size_t avail = r->end - r->pos;
    const void* sp = memchr(p, ' ', avail);
    if (sp) {
      const size_t n = (const unsigned char*)sp - p;
      key.append((const char*)p, n);
      r->pos += n + 1;
      break;
    }
    key.append((const char*)p, avail);
    r->pos += avail;
    if (key.size() > 4096) return fail(r, "key longer than 4096 bytes: not a Kaldi ark");
  }
  // strip leading whitespace/newlines that may separate records
  size_t a = 0;
  while (a < key.size() && (key[a] == '\n' || key[a] == '\r' || key[a] == '\t')) ++a;
  key.erase(0, a);
  if (key.empty()) return fail(r, "empty key");
  return parse_matrix_header(r, key);
}

// scp mode: position the descriptor at `offset` of the current file; bytes already in the read-ahead window are kept
int seek_to(xv_ark_reader* r, int64_t offset) {
  if (r->map) {
    if (offset < 0 || (uint64_t)offset > r->map_len) return fail(r, "scp offset beyond the end of the ark");
    r->pos = (size_t)offset;
    return XV_OK;
  }
  const int64_t cur = r->file_pos - (int64_t)(r->end - r->pos);
  if (offset >= cur && offset <= r->file_pos) {
    r->pos += (size_t)(offset - cur);
    return XV_OK;
  }
  if (lseek(r->fd, (off_t)offset, SEEK_SET) < 0) return fail(r, "lseek failed (scp entries need a seekable file)");
  r->pos = r->end = 0;
  r->file_pos = offset;
  r->eof = false;
  return XV_OK;
}

// scp mode: header of the next table entry -> pending_*; 1 = parsed, 0 = end of table, <0 error
int parse_scp_entry(xv_ark_reader* r) {
  if (r->next_entry >= r->entries.size()) return 0;
  const xv_ark_reader::ScpEntry& e = r->entries[r->next_entry++];
  if (e.path != r->cur_path) {
    unmap_file(r);
    if (r->fd >= 0) close(r->fd);
    r->fd = open(r->paths[e.path].c_str(), O_RDONLY);
    if (r->fd < 0) {
      r->err = "cannot open " + r->paths[e.path] + ": " + strerror(errno);
      return XV_ERR_INVALID;
    }
    r->cur_path = e.path;
    r->pos = r->end = 0;
    r->file_pos = 0;
    r->eof = false;
    map_file(r, false);
  }
  const int rc = seek_to(r, e.offset);
  if (rc < 0) return rc;
  std::string key = e.key;
  return parse_matrix_header(r, key);
}

// payload of the pending record -> dst (float32 row-major) or skipped when dst == nullptr
int read_payload(xv_ark_reader* r, float* dst) {
  const int64_t rows = r->pending_rows, cols = r->pending_cols, n = rows * cols;
  r->have_pending = false;
  if (r->pending_kind == 1 || r->pending_kind == 2) {
    const size_t es = r->pending_kind == 1 ? 4 : 8;
    if (es == 4 && dst && r->map) {         // mapped file: note the copy, do it with the rest of the batch
      const size_t need = (size_t)n * 4;
      if (r->end - r->pos < need) return fail(r, "truncated matrix payload");
      r->copies.push_back({(int64_t)r->pos, dst, need});
      r->pos += need;
      return XV_OK;
    }
    if (es == 4 && dst) {
      // float payload: take what the read-ahead window already holds, then read() the rest straight
      // into the destination (one copy instead of two)
      size_t need = (size_t)n * 4, have = r->end - r->pos;
      if (have > need) have = need;
      memcpy(dst, r->win + r->pos, have);
      r->pos += have;
      unsigned char* out = reinterpret_cast<unsigned char*>(dst) + have;
      need -= have;
      if (need > 0 && r->map) return fail(r, "truncated matrix payload");
      while (need > 0) {
        const ssize_t got = read(r->fd, out, need);
        if (got < 0) {
          if (errno == EINTR) continue;
          return fail(r, "read failed inside a matrix payload");
        }
        if (got == 0) { r->eof = true; return fail(r, "truncated matrix payload"); }
        out += got;
        need -= (size_t)got;
        r->file_pos += got;
      }
      return XV_OK;
    }
    int64_t left = n;
    float* out = dst;
    while (left > 0) {                       // stream through the window: a record may exceed it
      const int64_t want = left < (int64_t)(kChunk / es) ? left : (int64_t)(kChunk / es);
      if (!fill(r, (size_t)want * es)) return fail(r, "truncated matrix payload");
      const unsigned char* q = r->win + r->pos;
      if (out) {
        if (es == 4) {
          memcpy(out, q, (size_t)want * 4);
        } else {
          for (int64_t i = 0; i < want; ++i) { double d; memcpy(&d, q + 8 * i, 8); out[i] = (float)d; }
        }
        out += want;
      }
      r->pos += (size_t)want * es;
      left -= want;
    }
    return XV_OK;
  }
  // 'CM ': per-column percentile headers then column-major bytes (dataset/kaldi_io.py:1071-1115); the
  // arithmetic is done in double and rounded once, like Kaldi's CompressedMatrix
  const size_t total = (size_t)cols * 8 + (size_t)n;
  if (!fill(r, total)) return fail(r, "truncated compressed-matrix payload");
  const unsigned char* q = r->win + r->pos;
  if (dst) {
    const double gmin = r->pending_min, grange = r->pending_range;
    const unsigned char* data = q + (size_t)cols * 8;
    for (int64_t c = 0; c < cols; ++c) {
      uint16_t h[4];
      memcpy(h, q + 8 * c, 8);
      double p[4];
      for (int i = 0; i < 4; ++i) p[i] = (double)(float)(gmin + grange * 1.52590218966964e-05 * h[i]);
      const unsigned char* col = data + (size_t)c * rows;
      for (int64_t t = 0; t < rows; ++t) {
        const int v = col[t];
        double x;
        if (v <= 64) x = p[0] + (p[1] - p[0]) / 64. * v;
        else if (v <= 192) x = p[1] + (p[2] - p[1]) / 128. * (v - 64);
        else x = p[2] + (p[3] - p[2]) / 63. * (v - 192);
        dst[t * cols + c] = (float)x;
      }
    }
  }
  r->pos += total;
  return XV_OK;
}

}  // namespace

extern "C" {

int xv_ark_open(const char* path, int fd, xv_ark_reader** out) {
  if (!out) return XV_ERR_INVALID;
  *out = nullptr;
  xv_ark_reader* r = new (std::nothrow) xv_ark_reader();
  if (!r) return XV_ERR_HIP;
  if (path) {
    r->fd = open(path, O_RDONLY);
    if (r->fd < 0) { delete r; return XV_ERR_INVALID; }
    r->own_fd = true;
  } else {
    if (fd < 0) { delete r; return XV_ERR_INVALID; }
    r->fd = fd;
  }
  r->buf.resize(kChunk);
  r->win = r->buf.data();
  if (path) map_file(r, true);
  r->copy_threads = default_copy_threads();
  *out = r;
  return XV_OK;
}

int xv_ark_open_scp(const char* scp_path, xv_ark_reader** out) {
  if (!out || !scp_path) return XV_ERR_INVALID;
  *out = nullptr;
  FILE* f = fopen(scp_path, "r");
  if (!f) return XV_ERR_INVALID;
  xv_ark_reader* r = new (std::nothrow) xv_ark_reader();
  if (!r) { fclose(f); return XV_ERR_HIP; }
  r->scp = true;
  r->own_fd = true;
  r->copy_threads = default_copy_threads();
  r->buf.resize(kChunk);
  r->win = r->buf.data();
  char* line = nullptr;
  size_t cap = 0;
  ssize_t len;
  bool bad = false;
  while ((len = getline(&line, &cap, f)) >= 0) {
    while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r' || line[len - 1] == ' ' || line[len - 1] == '\t')) line[--len] = 0;
    char* p = line;
    while (*p == ' ' || *p == '\t') ++p;
    if (!*p) continue;
    char* sp = p;
    while (*sp && *sp != ' ' && *sp != '\t') ++sp;
    if (!*sp) { bad = true; break; }                       // a key without an rxfilename
    std::string key(p, sp - p);
    while (*sp == ' ' || *sp == '\t') ++sp;
    std::string rx(sp);
    if (rx.empty() || rx.back() == ']' || rx.back() == '|') { bad = true; break; }   // ranges / pipes: not handled natively
    int64_t offset = 0;
    const size_t colon = rx.rfind(':');
    if (colon != std::string::npos && colon + 1 < rx.size() &&
        rx.find_first_not_of("0123456789", colon + 1) == std::string::npos) {
      offset = strtoll(rx.c_str() + colon + 1, nullptr, 10);
      rx.erase(colon);
    }
    int pi = -1;
    if (!r->paths.empty() && r->paths.back() == rx) pi = (int)r->paths.size() - 1;     // consecutive entries share arks
    if (pi < 0)
      for (size_t i = 0; i < r->paths.size(); ++i) if (r->paths[i] == rx) { pi = (int)i; break; }
    if (pi < 0) { r->paths.push_back(rx); pi = (int)r->paths.size() - 1; }
    r->entries.push_back({std::move(key), pi, offset});
  }
  free(line);
  fclose(f);
  if (bad) { delete r; return XV_ERR_UNSUPPORTED; }
  *out = r;
  return XV_OK;
}

int64_t xv_ark_scp_count(const xv_ark_reader* r) { return (r && r->scp) ? (int64_t)r->entries.size() : XV_ERR_INVALID; }

int xv_ark_scp_shapes(xv_ark_reader* r, int32_t* rows, int32_t* cols, int64_t capacity) {
  if (!r || !r->scp || !rows || !cols || capacity < (int64_t)r->entries.size()) return XV_ERR_INVALID;
  r->err.clear();
  r->next_entry = 0;
  r->have_pending = false;
  int rc = XV_OK;
  for (size_t i = 0; i < r->entries.size(); ++i) {
    rc = parse_scp_entry(r);
    if (rc <= 0) { rc = rc == 0 ? XV_ERR_INVALID : rc; break; }
    rows[i] = r->pending_rows;
    cols[i] = r->pending_cols;
    r->have_pending = false;
    rc = XV_OK;
  }
  r->next_entry = 0;                     // rewind: the table can be read after it has been measured
  r->have_pending = false;
  return rc;
}

int xv_ark_next_batch(xv_ark_reader* r, int64_t max_frames, int max_utts, int min_frames, float* dst,
                      int64_t dst_capacity, int32_t* offsets, char* keys, int64_t keys_capacity, int* n_utts,
                      int* dim) {
  if (!r || !dst || !offsets || !keys || !n_utts || !dim || max_utts < 1) return XV_ERR_INVALID;
  r->err.clear();
  r->copies.clear();
  int n = 0, d = -1;
  int64_t frames = 0, kpos = 0;
  offsets[0] = 0;
  while (n < max_utts && frames < max_frames) {
    if (!r->have_pending) {
      if (r->scp && r->map && r->next_entry < r->entries.size() && r->entries[r->next_entry].path != r->cur_path)
        run_copies(r);                                // the next entry lives in another ark: this mapping goes away
      const int rc = r->scp ? parse_scp_entry(r) : parse_header(r);
      if (rc < 0) return rc;
      if (rc == 0) break;
    }
    const int64_t rows = r->pending_rows, cols = r->pending_cols;
    if (rows < min_frames) {                          // extract.py:65-67: too short, skipped
      const int rc = read_payload(r, nullptr);
      if (rc < 0) return rc;
      r->skipped_short++;
      continue;
    }
    if (d < 0) d = (int)cols;
    const bool fits = cols == d && (frames + rows) * d <= dst_capacity &&
                      kpos + (int64_t)r->pending_key.size() + 1 <= keys_capacity && frames + rows <= INT32_MAX;
    if (!fits) {
      if (n == 0) {
        if (cols != d) return fail(r, "feature dimension changed inside the ark");
        return fail(r, "a single utterance does not fit the destination buffer");
      }
      break;                                          // keep the header; deliver it with the next batch
    }
    memcpy(keys + kpos, r->pending_key.data(), r->pending_key.size());
    kpos += (int64_t)r->pending_key.size();
    keys[kpos++] = '\n';
    const int rc = read_payload(r, dst + frames * d);
    if (rc < 0) return rc;
    frames += rows;
    offsets[++n] = (int32_t)frames;
  }
  run_copies(r);
  if (r->copy_failed.exchange(false)) { return fail(r, "pread failed inside a matrix payload"); }
  *n_utts = n;
  *dim = d < 0 ? 0 : d;
  return n;
}

int xv_ark_pending_shape(const xv_ark_reader* r, int32_t* rows, int32_t* cols) {
  if (!r || !rows || !cols || !r->have_pending) return XV_ERR_STATE;
  *rows = r->pending_rows;
  *cols = r->pending_cols;
  return XV_OK;
}

int64_t xv_ark_skipped(const xv_ark_reader* r) { return r ? r->skipped_short : 0; }

int xv_ark_set_copy_threads(xv_ark_reader* r, int n) {
  if (!r) return XV_ERR_INVALID;
  r->copy_threads = n < 1 ? 1 : (n > 16 ? 16 : n);
  return XV_OK;
}

const char* xv_ark_error(const xv_ark_reader* r) { return r ? r->err.c_str() : "null reader"; }

void xv_ark_close(xv_ark_reader* r) {
  if (!r) return;
  unmap_file(r);
  if (r->own_fd && r->fd >= 0) close(r->fd);
  delete r;
}

int64_t xv_ark_format_vectors(const char* keys, int n, const float* data, int dim, int64_t ld, char* out,
                              int64_t out_capacity) {
  if (!keys || !data || !out || n < 0 || dim < 0) return XV_ERR_INVALID;
  int64_t pos = 0;
  const char* k = keys;
  for (int i = 0; i < n; ++i) {
    const char* e = strchr(k, '\n');
    if (!e) return XV_ERR_INVALID;
    const int64_t klen = e - k;
    const int64_t need = klen + 1 + 2 + 3 + 1 + 4 + (int64_t)dim * 4;
    if (pos + need > out_capacity) return XV_ERR_WORKSPACE;
    memcpy(out + pos, k, (size_t)klen);
    pos += klen;
    out[pos++] = ' ';
    memcpy(out + pos, "\0BFV \4", 6);
    pos += 6;
    const int32_t d32 = dim;
    memcpy(out + pos, &d32, 4);
    pos += 4;
    memcpy(out + pos, data + (int64_t)i * ld, (size_t)dim * 4);
    pos += (int64_t)dim * 4;
    k = e + 1;
  }
  return pos;
}

// Pack n row blocks (src[i], nbytes[i]) back to back into dst with a few threads: the staging copy of Trainer.submit_list (a ragged
// batch arrives as a list of separate [T_i, d] matrices; 9 MB per 256 x 300 frames, memcpy-bound on one core).  Returns the bytes copied.
int64_t xv_pack_rows(const void* const* src, const int64_t* nbytes, int n, void* dst, int threads) {
  if (!src || !nbytes || !dst || n < 0) return XV_ERR_INVALID;
  std::vector<int64_t> start((size_t)n + 1, 0);
  for (int i = 0; i < n; ++i) {
    if (nbytes[i] < 0 || (!src[i] && nbytes[i] > 0)) return XV_ERR_INVALID;
    start[i + 1] = start[i] + nbytes[i];
  }
  const int64_t total = start[n];
  int nt = threads < 1 ? 1 : (threads > 16 ? 16 : threads);
  if (total < ((int64_t)1 << 20)) nt = 1;
  auto work = [&](int64_t lo, int64_t hi) {            // bytes [lo, hi) of the concatenation
    int i = (int)(std::upper_bound(start.begin(), start.end(), lo) - start.begin()) - 1;
    while (lo < hi && i < n) {
      const int64_t a = lo - start[i];
      const int64_t len = std::min(hi, start[i + 1]) - lo;
      if (len > 0) memcpy(static_cast<char*>(dst) + lo, static_cast<const char*>(src[i]) + a, (size_t)len);
      lo += len > 0 ? len : 0;
      ++i;
    }
  };
  if (nt == 1) {
    work(0, total);
  } else {
    std::vector<std::thread> th;
    const int64_t share = ((total + nt - 1) / nt + 63) & ~(int64_t)63;
    for (int t = 1; t < nt; ++t) {
      const int64_t lo = share * t, hi = std::min(lo + share, total);
      if (lo < hi) th.emplace_back(work, lo, hi);
    }
    work(0, std::min(share, total));
    for (auto& x : th) x.join();
  }
  return total;
}

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) of `n` bytes, continuing from `crc` (0 to start): the checksum of the blocks
// of a TF checkpoint-V2 index (LevelDB table format) and of its tensors (BundleEntryProto.crc32c); tf_checkpoint.py masks it.
// SSE4.2 has the instruction; the table form is the fallback.
namespace {
uint32_t crc32c_table[8][256];
bool crc32c_table_ready = false;
void crc32c_init() {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1) ? 0x82F63B78u : 0u);
    crc32c_table[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t) crc32c_table[t][i] = (crc32c_table[t - 1][i] >> 8) ^ crc32c_table[0][crc32c_table[t - 1][i] & 0xff];
  crc32c_table_ready = true;
}
#if defined(__x86_64__)
__attribute__((target("sse4.2"))) uint32_t crc32c_hw(uint32_t c, const unsigned char* p, size_t n) {
  uint64_t c64 = c;
  while (n >= 8) {
    uint64_t v;
    memcpy(&v, p, 8);
    c64 = __builtin_ia32_crc32di(c64, v);
    p += 8;
    n -= 8;
  }
  c = (uint32_t)c64;
  while (n--) c = __builtin_ia32_crc32qi(c, *p++);
  return c;
}
#endif
}  // namespace

uint32_t xv_crc32c(uint32_t crc, const void* data, int64_t n) {
  if (!data || n <= 0) return crc;
  const unsigned char* p = static_cast<const unsigned char*>(data);
  uint32_t c = ~crc;
#if defined(__x86_64__)
  if (__builtin_cpu_supports("sse4.2")) return ~crc32c_hw(c, p, (size_t)n);
#endif
  static std::once_flag once;
  std::call_once(once, crc32c_init);
  size_t len = (size_t)n;
  while (len >= 8) {                       // slicing-by-8
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = crc32c_table[7][lo & 0xff] ^ crc32c_table[6][(lo >> 8) & 0xff] ^ crc32c_table[5][(lo >> 16) & 0xff] ^ crc32c_table[4][lo >> 24] ^
        crc32c_table[3][hi & 0xff] ^ crc32c_table[2][(hi >> 8) & 0xff] ^ crc32c_table[1][(hi >> 16) & 0xff] ^ crc32c_table[0][hi >> 24];
    p += 8;
    len -= 8;
  }
  while (len--) c = (c >> 8) ^ crc32c_table[0][(c ^ *p++) & 0xff];
  return ~c;
}

}  // extern "C"
