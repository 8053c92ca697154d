/* dbfloor.c -- what limits the flow database's insert rate (run ON the GPU box; built by tools/dbprobe/dbfloor.sh).
 *
 * One frame1 of the analysis = one `keypoints` row + eight `optical_flow` rows with three blobs each, in one transaction
 * (reference cpp/database.cc:183-214 WriteKeypoints / WriteImagePairFlow, cpp/opticalflow.cc:149-151).  Modes:
 *   product      what csrc/host/flow_database.cc does: 64-KiB pages, synchronous OFF, rollback journal (TRUNCATE) while
 *                loading, sqlite3_bind_blob(SQLITE_STATIC) of the whole blob
 *   zeroblob     INSERT with zeroblob(n), then sqlite3_blob_open + sqlite3_blob_write (incremental I/O: no bind copy)
 *   mmap         product + PRAGMA mmap_size = 16 GiB
 *   one_txn      product with ONE transaction for the whole clip
 *   page4k       product with SQLite's default page size
 *   raw_write    the same bytes with write(2) in one call per blob into a plain file (the file system's floor)
 *   memcpy       the same bytes copied once in memory
 * Usage: dbfloor <directory> <keypoints per frame> <frames>
 */
#include <fcntl.h>
#include <sqlite3.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

static double now(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}
#define SQL(db, s) do { char* e_ = 0; if (sqlite3_exec(db, s, 0, 0, &e_) != SQLITE_OK) { fprintf(stderr, "%s: %s\n", s, e_); exit(1); } } while (0)

static const char* SCHEMA =
    "CREATE TABLE IF NOT EXISTS keypoints(image_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL, keypoints BLOB NOT NULL);"
    "CREATE TABLE IF NOT EXISTS optical_flow(image_id_from INTEGER NOT NULL, image_id_to INTEGER NOT NULL, rows INTEGER NOT NULL,"
    " src_keypoints_indices BLOB NOT NULL, tgt_keypoints BLOB NOT NULL, flow_errors BLOB NOT NULL,"
    " PRIMARY KEY(image_id_from, image_id_to), FOREIGN KEY(image_id_from) REFERENCES keypoints(image_id) ON DELETE CASCADE);";
static const int SKIPS[8] = {-8, -4, -2, -1, 1, 2, 4, 8};

static void rm(const char* path) {
    char b[512];
    unlink(path);
    snprintf(b, sizeof b, "%s-journal", path); unlink(b);
    snprintf(b, sizeof b, "%s-wal", path); unlink(b);
    snprintf(b, sizeof b, "%s-shm", path); unlink(b);
}

static double run_sqlite(const char* dir, const char* mode, int n, int frames, const uint8_t* buf) {
    char path[512];
    snprintf(path, sizeof path, "%s/dbfloor_%s.db", dir, mode);
    rm(path);
    sqlite3* db;
    if (sqlite3_open_v2(path, &db, SQLITE_OPEN_READWRITE | SQLITE_OPEN_CREATE | SQLITE_OPEN_NOMUTEX, 0) != SQLITE_OK) exit(1);
    const int zero = !strcmp(mode, "zeroblob"), one = !strcmp(mode, "one_txn");
    if (strcmp(mode, "page4k")) SQL(db, "PRAGMA page_size=65536");
    SQL(db, "PRAGMA synchronous=OFF");
    SQL(db, "PRAGMA journal_mode=TRUNCATE");
    SQL(db, "PRAGMA temp_store=MEMORY");
    SQL(db, "PRAGMA foreign_keys=ON");
    SQL(db, "PRAGMA auto_vacuum=1");
    if (!strcmp(mode, "mmap")) SQL(db, "PRAGMA mmap_size=17179869184");
    SQL(db, SCHEMA);
    sqlite3_stmt *kp, *fl;
    sqlite3_prepare_v2(db, zero ? "INSERT INTO keypoints(image_id, rows, keypoints) VALUES(?, ?, zeroblob(?))"
                                : "INSERT INTO keypoints(image_id, rows, keypoints) VALUES(?, ?, ?)", -1, &kp, 0);
    sqlite3_prepare_v2(db, zero ? "INSERT INTO optical_flow(image_id_from, image_id_to, rows, src_keypoints_indices, tgt_keypoints, flow_errors) VALUES(?, ?, ?, zeroblob(?), zeroblob(?), zeroblob(?))"
                                : "INSERT INTO optical_flow(image_id_from, image_id_to, rows, src_keypoints_indices, tgt_keypoints, flow_errors) VALUES(?, ?, ?, ?, ?, ?)", -1, &fl, 0);
    const double t0 = now();
    if (one) SQL(db, "BEGIN");
    for (int f = 1; f <= frames; f++) {
        if (!one) SQL(db, "BEGIN");
        sqlite3_bind_int(kp, 1, f);
        sqlite3_bind_int(kp, 2, n);
        if (zero) sqlite3_bind_int(kp, 3, n * 8); else sqlite3_bind_blob(kp, 3, buf, n * 8, SQLITE_STATIC);
        if (sqlite3_step(kp) != SQLITE_DONE) { fprintf(stderr, "kp: %s\n", sqlite3_errmsg(db)); exit(1); }
        sqlite3_reset(kp);
        if (zero) {
            sqlite3_blob* b;
            sqlite3_blob_open(db, "main", "keypoints", "keypoints", sqlite3_last_insert_rowid(db), 1, &b);
            sqlite3_blob_write(b, buf, n * 8, 0);
            sqlite3_blob_close(b);
        }
        for (int k = 0; k < 8; k++) {
            sqlite3_bind_int(fl, 1, f);
            sqlite3_bind_int(fl, 2, f + SKIPS[k]);
            sqlite3_bind_int(fl, 3, n);
            if (zero) {
                sqlite3_bind_int(fl, 4, n * 4); sqlite3_bind_int(fl, 5, n * 8); sqlite3_bind_int(fl, 6, n * 4);
            } else {
                sqlite3_bind_blob(fl, 4, buf, n * 4, SQLITE_STATIC);
                sqlite3_bind_blob(fl, 5, buf, n * 8, SQLITE_STATIC);
                sqlite3_bind_blob(fl, 6, buf, n * 4, SQLITE_STATIC);
            }
            if (sqlite3_step(fl) != SQLITE_DONE) { fprintf(stderr, "fl: %s\n", sqlite3_errmsg(db)); exit(1); }
            sqlite3_reset(fl);
            if (zero) {
                const sqlite3_int64 rid = sqlite3_last_insert_rowid(db);
                const char* cols[3] = {"src_keypoints_indices", "tgt_keypoints", "flow_errors"};
                const int sz[3] = {n * 4, n * 8, n * 4};
                for (int c = 0; c < 3; c++) {
                    sqlite3_blob* b;
                    if (sqlite3_blob_open(db, "main", "optical_flow", cols[c], rid, 1, &b) != SQLITE_OK) { fprintf(stderr, "blob_open: %s\n", sqlite3_errmsg(db)); exit(1); }
                    sqlite3_blob_write(b, buf, sz[c], 0);
                    sqlite3_blob_close(b);
                }
            }
        }
        if (!one) SQL(db, "COMMIT");
    }
    if (one) SQL(db, "COMMIT");
    SQL(db, "PRAGMA journal_mode=WAL");
    sqlite3_finalize(kp);
    sqlite3_finalize(fl);
    sqlite3_close(db);
    const double dt = now() - t0;
    rm(path);
    return dt;
}

int main(int argc, char** argv) {
    const char* dir = argc > 1 ? argv[1] : "/tmp";
    const int n = argc > 2 ? atoi(argv[2]) : 40600, frames = argc > 3 ? atoi(argv[3]) : 120;
    const size_t per_frame = (size_t)n * 8 + 8 * (size_t)n * 16;
    uint8_t* buf = malloc((size_t)n * 8);
    for (size_t i = 0; i < (size_t)n * 8; i++) buf[i] = (uint8_t)(i * 2654435761u >> 13);
    printf("{\"sqlite\": \"%s\", \"directory\": \"%s\", \"keypoints\": %d, \"frames\": %d, \"bytes_per_frame\": %zu", sqlite3_libversion(), dir, n, frames, per_frame);
    const char* modes[] = {"product", "zeroblob", "mmap", "one_txn", "page4k"};
    for (int m = 0; m < 5; m++) {
        const double dt = run_sqlite(dir, modes[m], n, frames, buf);
        printf(", \"%s\": {\"frames_per_s\": %.1f, \"GB_per_s\": %.3f}", modes[m], frames / dt, per_frame * frames / dt / 1e9);
        fflush(stdout);
    }
    {   /* raw write */
        char path[512];
        snprintf(path, sizeof path, "%s/dbfloor_raw.bin", dir);
        const int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644);
        const double t0 = now();
        for (int f = 0; f < frames; f++) {
            if (write(fd, buf, (size_t)n * 8) < 0) return 1;
            for (int k = 0; k < 8; k++)
                if (write(fd, buf, (size_t)n * 4) < 0 || write(fd, buf, (size_t)n * 8) < 0 || write(fd, buf, (size_t)n * 4) < 0) return 1;
        }
        close(fd);
        const double dt = now() - t0;
        unlink(path);
        printf(", \"raw_write\": {\"frames_per_s\": %.1f, \"GB_per_s\": %.3f}", frames / dt, per_frame * frames / dt / 1e9);
    }
    {
        uint8_t* dst = malloc(per_frame);
        const double t0 = now();
        for (int f = 0; f < frames; f++) {
            size_t o = 0;
            memcpy(dst + o, buf, (size_t)n * 8); o += (size_t)n * 8;
            for (int k = 0; k < 8; k++) {
                memcpy(dst + o, buf, (size_t)n * 4); o += (size_t)n * 4;
                memcpy(dst + o, buf, (size_t)n * 8); o += (size_t)n * 8;
                memcpy(dst + o, buf, (size_t)n * 4); o += (size_t)n * 4;
            }
        }
        const double dt = now() - t0;
        printf(", \"memcpy\": {\"frames_per_s\": %.1f, \"GB_per_s\": %.3f}", frames / dt, per_frame * frames / dt / 1e9);
        free(dst);
    }
    printf("}\n");
    return 0;
}
