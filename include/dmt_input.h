/* dmt_input.h -- C ABI of libdmt_input.so: the host-side input stage of the DMT train step.
 *
 * Replaces, for the reference's data feed (DMT_code/data_feed/tfrecord_mask.py:23-84,120-158 parse_single_line + TFRecordDataset,
 * DMT_code/data_feed/index_tables.py:8-45 LookupTables.transform_id2index), what TensorFlow 1.12's C++ runtime does there:
 *   - TFRecord framing with masked CRC32C                        (tf.data.TFRecordDataset)
 *   - tf.Example protobuf wire decode                             (tf.parse_example with VarLenFeature / FixedLenFeature)
 *   - string id -> index with FarmHash Fingerprint64 OOV buckets  (tf.contrib.lookup.index_table_from_tensor(mapping,
 *                                                                  num_oov_buckets = id_size - len(mapping), default_value = 0))
 *   - left-aligned zero-padded [B, T] columns + lengths           (tf.sparse_tensor_to_dense / reduce_sum of ones,
 *                                                                  mmoe_transformer.py:135-142)
 * Plain C, host pointers, caller-owned output buffers, no global mutable state besides a thread-local error string.
 * All functions return 0 (DMT_IN_OK) on success and a negative code on failure unless stated otherwise.
 */
#ifndef DMT_INPUT_H
#define DMT_INPUT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMT_IN_OK 0
#define DMT_IN_ERR_ARG (-1)
#define DMT_IN_ERR_IO (-2)
#define DMT_IN_ERR_FORMAT (-3)
#define DMT_IN_ERR_RANGE (-4)

const char* dmt_input_last_error(void);
int32_t dmt_input_version(void);

/* ---- checksums / hashes ------------------------------------------------------------------------------------------------
 * CRC-32C (Castagnoli, reflected 0x82F63B78) and TFRecord's mask ((crc >> 15 | crc << 17) + 0xa282ead8).
 * Fingerprint64 = farmhashna::Hash64 of Google FarmHash: the hash of tf.string_to_hash_bucket_fast. */
uint32_t dmt_crc32c(const void* data, uint64_t n);
uint32_t dmt_masked_crc32c(const void* data, uint64_t n);
uint64_t dmt_fingerprint64(const void* data, uint64_t n);

/* ---- TFRecord reader ---------------------------------------------------------------------------------------------------
 * record := u64 length | u32 masked_crc32c(length) | payload | u32 masked_crc32c(payload)      (little endian)
 * dmt_tfrecord_next: 1 = a record (payload valid until the next call on this reader), 0 = end of file, < 0 = error. */
typedef struct dmt_tfrecord_reader dmt_tfrecord_reader;
int dmt_tfrecord_open(const char* path, int32_t verify_crc, dmt_tfrecord_reader** out);
int dmt_tfrecord_next(dmt_tfrecord_reader* r, const uint8_t** payload, uint64_t* len);
void dmt_tfrecord_close(dmt_tfrecord_reader* r);

/* ---- vocabulary --------------------------------------------------------------------------------------------------------
 * keys[i] (key_lens[i] bytes) -> i; the first occurrence of a duplicated key wins.  An id that is not a key maps to
 * n_keys + Fingerprint64(id) % (id_size - n_keys), or to 0 when id_size == n_keys (no OOV buckets: the Time* tables). */
typedef struct dmt_vocab dmt_vocab;
int dmt_vocab_create(const char* const* keys, const uint32_t* key_lens, int64_t n_keys, int64_t id_size, dmt_vocab** out);
int64_t dmt_vocab_lookup(const dmt_vocab* v, const void* id, uint64_t n);
void dmt_vocab_destroy(dmt_vocab* v);

/* ---- batch parser ------------------------------------------------------------------------------------------------------
 * One call turns B serialized tf.Example payloads into the padded columns the device batch is built from.
 *   id feature   (vocab != NULL): Example key `name` is a bytes list of ids -> idx[b, 0:len] = lookup(id), lens[b] = len;
 *                                 Example key `name` + "Wts" (float list, optional) -> wts[b, 0:len] (when wts != NULL).
 *   float feature (vocab == NULL): Example key `name` is a float list of exactly max_len values -> dense[b, :]
 *                                 (`features` 615, `mask` 5, `label` 1); a missing key leaves zeros.
 * Rows are zero filled first.  A list longer than max_len is an error (DMT_IN_ERR_RANGE), like a shape mismatch in
 * tf.parse_example.  n_threads <= 1: the calling thread does all the work; otherwise the rows are split over n_threads workers of a
 * process-wide pool of parked threads (created on first use, one job at a time: concurrent callers take turns). */
typedef struct {
  const char* name;
  const dmt_vocab* vocab;
  int32_t max_len;
  int32_t* idx;    /* [B, max_len] */
  float* wts;      /* [B, max_len] or NULL */
  int32_t* lens;   /* [B] */
  float* dense;    /* [B, max_len] (float features) */
  int32_t* n_wts_not_one;   /* optional: incremented (atomically) by the number of parsed weights != 1.0 -- 0 at the end means the
                               feature is an unweighted mean and the caller may drop the weights column */
} dmt_feature_spec;

int dmt_parse_batch(const uint8_t* const* payloads, const uint64_t* payload_lens, int32_t B, const dmt_feature_spec* feats,
                    int32_t n_feats, int32_t n_threads);

/* The same straight off a TFRecord file: reads up to B records from the reader and parses them; the payload crc check and the
 * decode run on the worker threads, the payloads never surface.  Returns the number of records parsed (rows [n, B) stay zero),
 * 0 at end of file, < 0 on error. */
int dmt_tfrecord_parse_batch(dmt_tfrecord_reader* r, int32_t B, const dmt_feature_spec* feats, int32_t n_feats, int32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif /* DMT_INPUT_H */
