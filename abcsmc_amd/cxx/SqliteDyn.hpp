// SqliteDyn.hpp -- RAII access to the system SQLite for the AbcSmc storage layer.
//
// The reference wraps SQLite in lib/sqdb (Db / Statement / Convertor, /root/reference/lib/sqdb/include/sqdb.h:126-206)
// and ships sqlite3.h without the amalgamation.  This image has the runtime library (libsqlite3.so.0) but no
// header, so the handful of C entry points used here are declared locally and resolved with dlopen at first use.
// Kept from sqdb: a statement's next() waits and retries while the database is locked by another process
// (lib/sqdb/src/sqdb.cpp:271-290); here compiling a statement does the same (it needs the schema, which an
// exclusive transaction of another worker also locks).  Any other failure throws.
#ifndef ABCSMC_AMD_SQLITEDYN_HPP
#define ABCSMC_AMD_SQLITEDYN_HPP

#include <dlfcn.h>
#include <unistd.h>

#include <cstdint>
#include <stdexcept>
#include <string>

namespace sqdyn {

struct sqlite3;
struct sqlite3_stmt;

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

struct Api {
    int (*open)(const char*, sqlite3**);
    int (*close)(sqlite3*);
    int (*prepare_v2)(sqlite3*, const char*, int, sqlite3_stmt**, const char**);
    int (*step)(sqlite3_stmt*);
    int (*reset)(sqlite3_stmt*);
    int (*finalize)(sqlite3_stmt*);
    int (*column_count)(sqlite3_stmt*);
    int (*column_type)(sqlite3_stmt*, int);
    long long (*column_int64)(sqlite3_stmt*, int);
    double (*column_double)(sqlite3_stmt*, int);
    const unsigned char* (*column_text)(sqlite3_stmt*, int);
    const char* (*errmsg)(sqlite3*);
    int (*busy_timeout)(sqlite3*, int);
    long long (*last_insert_rowid)(sqlite3*);

    static const Api& get() {
        static const Api api = load();
        return api;
    }

   private:
    template <typename F> static void sym(void* h, F& f, const char* name) {
        f = reinterpret_cast<F>(dlsym(h, name));
        if (!f) throw Error(-1, std::string("libsqlite3: missing symbol ") + name);
    }
    static Api load() {
        void* h = nullptr;
        for (const char* n : {"libsqlite3.so.0", "libsqlite3.so"}) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
        if (!h) throw Error(-1, "cannot load libsqlite3.so.0 (needed by the AbcSmc storage layer)");
        Api a;
        sym(h, a.open, "sqlite3_open");
        sym(h, a.close, "sqlite3_close");
        sym(h, a.prepare_v2, "sqlite3_prepare_v2");
        sym(h, a.step, "sqlite3_step");
        sym(h, a.reset, "sqlite3_reset");
        sym(h, a.finalize, "sqlite3_finalize");
        sym(h, a.column_count, "sqlite3_column_count");
        sym(h, a.column_type, "sqlite3_column_type");
        sym(h, a.column_int64, "sqlite3_column_int64");
        sym(h, a.column_double, "sqlite3_column_double");
        sym(h, a.column_text, "sqlite3_column_text");
        sym(h, a.errmsg, "sqlite3_errmsg");
        sym(h, a.busy_timeout, "sqlite3_busy_timeout");
        sym(h, a.last_insert_rowid, "sqlite3_last_insert_rowid");
        return a;
    }
};

constexpr unsigned kBusyWaitUs = 100000;   // the reference sleeps a whole second between tries
enum { SQLITE_OK_ = 0, SQLITE_BUSY_ = 5, SQLITE_ROW_ = 100, SQLITE_DONE_ = 101, SQLITE_NULL_ = 5 };

class Db;

class Stmt {
   public:
    Stmt(sqlite3* db, sqlite3_stmt* st) : db_(db), st_(st) {}
    Stmt(Stmt&& o) noexcept : db_(o.db_), st_(o.st_) { o.st_ = nullptr; }
    Stmt& operator=(Stmt&& o) noexcept {
        if (this != &o) { finish(); db_ = o.db_; st_ = o.st_; o.st_ = nullptr; }
        return *this;
    }
    Stmt(const Stmt&) = delete;
    Stmt& operator=(const Stmt&) = delete;
    ~Stmt() { finish(); }

    // true: a row is available; false: done.  Locked database: wait and retry.
    bool next() {
        for (;;) {
            const int rc = Api::get().step(st_);
            if (rc == SQLITE_ROW_) return true;
            if (rc == SQLITE_DONE_) return false;
            if (rc == SQLITE_BUSY_) { usleep(kBusyWaitUs); continue; }
            throw Error(rc, Api::get().errmsg(db_));
        }
    }
    int columns() const { return Api::get().column_count(st_); }
    bool is_null(int c) const { return Api::get().column_type(st_, c) == SQLITE_NULL_; }
    long long i64(int c) const { return Api::get().column_int64(st_, c); }
    double f64(int c) const { return Api::get().column_double(st_, c); }
    std::string text(int c) const {
        const unsigned char* t = Api::get().column_text(st_, c);
        return t ? std::string(reinterpret_cast<const char*>(t)) : std::string();
    }

   private:
    void finish() { if (st_) { Api::get().finalize(st_); st_ = nullptr; } }
    sqlite3* db_;
    sqlite3_stmt* st_;
};

class Db {
   public:
    explicit Db(const std::string& filename) : db_(nullptr) {
        const int rc = Api::get().open(filename.c_str(), &db_);
        if (rc != SQLITE_OK_) {
            const std::string m = db_ ? Api::get().errmsg(db_) : "out of memory";
            if (db_) Api::get().close(db_);
            throw Error(rc, "cannot open database " + filename + ": " + m);
        }
    }
    Db(const Db&) = delete;
    Db& operator=(const Db&) = delete;
    ~Db() { if (db_) Api::get().close(db_); }

    Stmt query(const std::string& sql) {
        sqlite3_stmt* st = nullptr;
        int rc;
        // compiling a statement reads the schema, which another worker's exclusive transaction also blocks
        while ((rc = Api::get().prepare_v2(db_, sql.c_str(), -1, &st, nullptr)) == SQLITE_BUSY_) usleep(kBusyWaitUs);
        if (rc != SQLITE_OK_) throw Error(rc, std::string(Api::get().errmsg(db_)) + " in: " + sql);
        return Stmt(db_, st);
    }
    // run a statement that returns no rows (or whose rows are not wanted)
    void exec(const std::string& sql) { Stmt s = query(sql); while (s.next()) {} }
    void begin_exclusive() { exec("BEGIN EXCLUSIVE;"); }
    void commit() { exec("COMMIT;"); }
    void rollback() { try { exec("ROLLBACK;"); } catch (const Error&) {} }
    long long last_insert_rowid() { return Api::get().last_insert_rowid(db_); }

    bool table_exists(const std::string& name) {
        Stmt s = query("select count(*) from sqlite_master where type='table' and name='" + name + "';");
        s.next();
        return s.i64(0) > 0;
    }

   private:
    sqlite3* db_;
};

}  // namespace sqdyn
#endif
