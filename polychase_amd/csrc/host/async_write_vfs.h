// async_write_vfs.h -- a SQLite VFS for the connection that bulk-loads the flow database: page writes to the main database
// file are copied out of SQLite's hands and carried out by worker threads.
//
// Why: one frame of the analysis is 5.5 MB of blobs (22 MB at 4K).  SQLite copies them into its pages (23 GB/s on the GPU
// box) and then hands every page to write(2), which on that box moves 10.9 GB/s per thread (page-cache allocation + copy,
// profiles/r03_dbfloor.json) -- the two in series are the 7.4 GB/s = 1340 frames/s ceiling the product call sat under while
// the GPU delivers 2990 frames/s.  write(2) to disjoint pages scales with threads; SQLite's pager is single-threaded.  With
// the writes taken off its thread the pager's cost per page is one more memcpy (67 GB/s).
//
// What it guarantees: the FILE is, at every point where anybody else can look at it, what it would be without the VFS --
// pending writes are carried out before a read of the file, before the size is asked for, before a sync / truncate / close
// and before the connection lets go of its write lock (xUnlock below RESERVED), i.e. before another connection or process
// can read.  Writes to one page keep their order (a page always goes to the same worker).  An error of a deferred write is
// reported by the next of those points (SQLITE_IOERR_WRITE): the COMMIT fails, like it would have.  Everything that is not
// a page write of the main database file (journal, WAL, shared memory, locks) goes straight to the default VFS.
// The reference opens its connection with the default VFS (cpp/database.cc:71-74); file format and content are identical
// (tests/test_core_cpu.py compares the files byte for byte).
#pragma once

// Registers the VFS (once per process, not as the default) and returns its name for sqlite3_open_v2, or nullptr if it cannot
// be registered -- the caller then opens with the default VFS.
const char* AsyncWriteVfsName();

struct AsyncWriteVfsCounters {
    unsigned long long deferred_writes = 0, deferred_bytes = 0, direct_writes = 0, drains = 0, drains_that_waited = 0, waits_for_a_slab = 0;
};
// process-wide totals since the start (tests, the stage report)
AsyncWriteVfsCounters AsyncWriteVfsTotals();
