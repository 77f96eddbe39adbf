/*
 * Readers / writers for the Dazzler read database and the DALIGNER overlap file (include/daccord_io.h).
 * Product host code, no device code.  The reference goes through libmaus2::dazzler::{db::DatabaseFile,
 * align::OverlapParser, align::DalignerIndexDecoder} (src/daccord.cpp:1328-1375, 2133-2160), which are not in the
 * reference tree; the byte layouts are those of DAZZ_DB (DB.h) and DALIGNER (align.h) as restated in SURVEY.md
 * section 10.  FORMAT UNPINNED: validated only by writer/reader round trips (tests/test_io_roundtrip.py).
 */
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <new>
#include <stdexcept>
#include <cstdlib>
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>
#include <sys/types.h>
#include "../../include/daccord_io.h"

namespace {

// x86-64 layouts of DAZZ_DB's structs (little endian)
enum { IDX_HEADER = 112, IDX_READ = 40 };
enum { DB_BEST = 0x800 };

static void put32(uint8_t * p, int32_t v) { std::memcpy(p,&v,4); }
static void put64(uint8_t * p, int64_t v) { std::memcpy(p,&v,8); }
static int32_t get32(uint8_t const * p) { int32_t v; std::memcpy(&v,p,4); return v; }
static int64_t get64(uint8_t const * p) { int64_t v; std::memcpy(&v,p,8); return v; }

static bool readFile(std::string const & fn, std::vector<uint8_t> & D, std::string & err)
{
	FILE * f = std::fopen(fn.c_str(),"rb");
	if ( !f ) { err = "cannot open " + fn; return false; }
	std::fseek(f,0,SEEK_END); long const n = std::ftell(f); std::fseek(f,0,SEEK_SET);
	D.resize(n < 0 ? 0 : n);
	bool const ok = D.empty() || std::fread(D.data(),1,D.size(),f) == D.size();
	std::fclose(f);
	if ( !ok ) err = "short read on " + fn;
	return ok;
}
static bool writeFile(std::string const & fn, void const * p, size_t n)
{
	FILE * f = std::fopen(fn.c_str(),"wb");
	if ( !f ) return false;
	bool const ok = (n == 0) || std::fwrite(p,1,n,f) == n;
	return (std::fclose(f) == 0) && ok;
}
// ".../foo.db" -> ".../.foo.<ext>"
static std::string hidden(std::string const & dbpath, char const * ext)
{
	size_t const slash = dbpath.find_last_of('/');
	std::string const dir = (slash == std::string::npos) ? std::string() : dbpath.substr(0,slash+1);
	std::string root = (slash == std::string::npos) ? dbpath : dbpath.substr(slash+1);
	if ( root.size() > 3 && root.compare(root.size()-3,3,".db") == 0 ) root.resize(root.size()-3);
	else if ( root.size() > 4 && root.compare(root.size()-4,4,".dam") == 0 ) root.resize(root.size()-4);
	return dir + "." + root + "." + ext;
}

}

struct dacc_db
{
	std::string err;
	std::vector<uint8_t> bps;
	std::vector<uint64_t> boff;
	std::vector<uint32_t> rlen;
};

struct dacc_las
{
	std::string err, path;
	int fd;
	int64_t novl; int32_t tspace; int32_t tbytes;
	uint64_t fsize;
	// index: byte offset of the first record of A read a (a = 0..maxaread+1; a read without records shares its
	// successor's offset), the reference's .las index (daccord.cpp:1075-1094, DalignerIndexDecoder) kept in memory
	std::vector<uint64_t> aoff;
	int64_t minaread, maxaread;
	bool index_from_sidecar;
	std::vector<uint8_t> buf;               // bytes of the requested range
	std::vector<dacc_pile> opiles; std::vector<dacc_overlap> oovl; std::vector<uint8_t> otrace;
	dacc_las() : fd(-1), novl(0), tspace(0), tbytes(1), fsize(0), minaread(0), maxaread(-1), index_from_sidecar(false) {}
	~dacc_las() { if ( fd >= 0 ) ::close(fd); }
};

namespace {

// sidecar index "<las>.daidx" (our own format, little endian): magic, file size, novl, tspace, minaread, maxaread, n, mtime of the
// .las (ns), FNV-1a of the .las header + its first and last 40-byte record header, aoff[n].  Written once (temp file + rename) by
// whichever process scans the .las first; -J g,G processes that start later load it instead of scanning the whole file again.
// A sidecar is accepted only for the very file it was made from: same size, record count and trace spacing, same modification
// time AND the same header / first record / last record bytes (a regenerated .las of the same size and count would otherwise
// reuse stale byte offsets).  Anything wrong with it -- including an absurd n -- means "scan again", never an error.
// DACC_LAS_INDEX=0 disables reading and writing it.
static char const IDXMAGIC[8] = {'D','A','C','C','I','D','X','2'};
enum { IDXHDR = 72 };
static bool sidecarEnabled() { char const * e = std::getenv("DACC_LAS_INDEX"); return !(e && e[0] == '0'); }
static bool preadAll(int fd, uint8_t * dst, uint64_t n, uint64_t off);
static uint64_t lasMtimeNs(dacc_las const & L)
{
	struct stat st;
	if ( ::fstat(L.fd,&st) != 0 ) return 0;
	return static_cast<uint64_t>(st.st_mtim.tv_sec)*1000000000ull + static_cast<uint64_t>(st.st_mtim.tv_nsec);
}
// FNV-1a over the 12 header bytes, the first record header and the 40 bytes at `lastoff` (the last record header)
static uint64_t lasFingerprint(dacc_las const & L, uint64_t const lastoff)
{
	uint8_t b[12+40+40]; std::memset(b,0,sizeof(b));
	if ( !preadAll(L.fd,b,12,0) ) return 0;
	if ( L.fsize >= 52 && !preadAll(L.fd,b+12,40,12) ) return 0;
	if ( lastoff >= 12 && lastoff + 40 <= L.fsize && !preadAll(L.fd,b+52,40,lastoff) ) return 0;
	uint64_t h = 1469598103934665603ull;
	for ( size_t i = 0; i < sizeof(b); ++i ) { h ^= b[i]; h *= 1099511628211ull; }
	return h ? h : 1;
}
static bool loadSidecar(dacc_las & L)
{
	FILE * f = std::fopen((L.path + ".daidx").c_str(),"rb");
	if ( !f ) return false;
	bool ok = false;
	try
	{
		uint8_t h[IDXHDR];
		ok = std::fread(h,1,sizeof(h),f) == sizeof(h) && std::memcmp(h,IDXMAGIC,8) == 0;
		if ( ok )
		{
			uint64_t const fs = get64(h+8); int64_t const novl = get64(h+16); int32_t const tsp = get32(h+24);
			int64_t const mina = get64(h+32), maxa = get64(h+40); uint64_t const n = get64(h+48);
			uint64_t const mt = get64(h+56), fp = get64(h+64);
			// (an index has one entry per A read id up to the last one that has a record, plus one: never more than records + 2)
			ok = fs == L.fsize && novl == L.novl && tsp == L.tspace && maxa >= -1 && n == static_cast<uint64_t>(maxa+2) && mt == lasMtimeNs(L) &&
			     (novl == 0 || n >= 2) && n <= (1ull<<32);
			if ( ok )
			{
				L.aoff.resize(n);
				ok = n == 0 || std::fread(L.aoff.data(),8,n,f) == n;
				for ( uint64_t i = 0; ok && i+1 < n; ++i ) ok = L.aoff[i] <= L.aoff[i+1];
				ok = ok && (n == 0 || (L.aoff[0] >= 12 && L.aoff[n-1] <= L.fsize));     // bytes behind the last record are ignored
				// the last record starts where the last A read's byte range holds its final record: found by walking that range
				if ( ok && n >= 2 )
				{
					uint64_t pos = L.aoff[n-2], last = pos; uint8_t r[4];
					while ( ok && pos + 40 <= L.aoff[n-1] )
					{
						ok = preadAll(L.fd,r,4,pos);
						int32_t const tlen = get32(r);
						if ( !ok || tlen < 0 ) { ok = false; break; }
						last = pos; pos += 40 + static_cast<uint64_t>(tlen)*L.tbytes;
					}
					ok = ok && pos == L.aoff[n-1] && fp == lasFingerprint(L,last);
				}
				if ( ok ) { L.minaread = mina; L.maxaread = maxa; }
			}
		}
	}
	catch ( std::exception const & ) { ok = false; }      // (a failed allocation for a corrupt n: scan instead)
	std::fclose(f);
	if ( !ok ) { L.aoff.clear(); L.aoff.shrink_to_fit(); }
	return ok;
}
static void writeSidecar(dacc_las const & L, uint64_t const lastrec)
{
	std::string const fn = L.path + ".daidx", tmp = fn + ".tmp." + std::to_string(static_cast<long long>(::getpid()));
	FILE * f = std::fopen(tmp.c_str(),"wb");
	if ( !f ) return;                      // read-only directory: every process scans for itself
	uint8_t h[IDXHDR]; std::memset(h,0,sizeof(h));
	std::memcpy(h,IDXMAGIC,8); put64(h+8,L.fsize); put64(h+16,L.novl); put32(h+24,L.tspace); put64(h+32,L.minaread); put64(h+40,L.maxaread); put64(h+48,L.aoff.size());
	put64(h+56,lasMtimeNs(L)); put64(h+64,lasFingerprint(L,lastrec));
	bool ok = std::fwrite(h,1,sizeof(h),f) == sizeof(h) && (L.aoff.empty() || std::fwrite(L.aoff.data(),8,L.aoff.size(),f) == L.aoff.size());
	ok = (std::fclose(f) == 0) && ok;
	if ( !ok || std::rename(tmp.c_str(),fn.c_str()) != 0 ) std::remove(tmp.c_str());
}
// all of [off,off+n) or false
static bool preadAll(int fd, uint8_t * dst, uint64_t n, uint64_t off)
{
	while ( n )
	{
		ssize_t const r = ::pread(fd,dst,n > (1ull<<30) ? (1ull<<30) : n,static_cast<off_t>(off));
		if ( r <= 0 ) return false;
		dst += r; off += r; n -= r;
	}
	return true;
}

}

extern "C" {

int dacc_db_open(const char * path, dacc_db ** out)
{
	if ( !path || !out ) return DACC_EINVAL;
	dacc_db * db = new dacc_db; *out = db;
	std::vector<uint8_t> idx, stub;
	if ( !readFile(path,stub,db->err) ) return DACC_EINVAL;                  // the stub must exist; its block table is not needed here
	if ( !readFile(hidden(path,"idx"),idx,db->err) ) return DACC_EINVAL;
	if ( idx.size() < IDX_HEADER ) { db->err = "index file too short"; return DACC_EINVAL; }
	int32_t const ureads = get32(idx.data()+0), cutoff = get32(idx.data()+8), allarr = get32(idx.data()+12);
	if ( ureads < 0 || idx.size() < static_cast<size_t>(IDX_HEADER) + static_cast<size_t>(ureads)*IDX_READ ) { db->err = "index file truncated"; return DACC_EINVAL; }
	std::vector<uint8_t> raw;
	if ( !readFile(hidden(path,"bps"),raw,db->err) ) return DACC_EINVAL;
	// trimmed view (DAZZ_DB Trim_DB / libmaus2 computeTrimVector): rlen >= cutoff and (all or best)
	for ( int32_t i = 0; i < ureads; ++i )
	{
		uint8_t const * r = idx.data() + IDX_HEADER + static_cast<size_t>(i)*IDX_READ;
		int32_t const rl = get32(r+4), flags = get32(r+32); int64_t const bo = get64(r+16);
		bool const keep = (rl >= cutoff) && ((allarr & 1) || (flags & DB_BEST) || cutoff < 0);
		if ( !keep ) continue;
		uint64_t const nb = (static_cast<uint64_t>(rl)+3)/4;
		if ( rl < 0 || bo < 0 || static_cast<uint64_t>(bo) + nb > raw.size() ) { db->err = "read payload outside the .bps file"; return DACC_EINVAL; }
		db->boff.push_back(db->bps.size()); db->rlen.push_back(rl);
		db->bps.insert(db->bps.end(),raw.begin()+bo,raw.begin()+bo+nb);
	}
	return DACC_OK;
}
void dacc_db_close(dacc_db * db) { delete db; }
const char * dacc_db_error(dacc_db * db) { return db ? db->err.c_str() : "null handle"; }
int dacc_db_arrays(dacc_db * db, const uint8_t ** bps, uint64_t * bps_bytes, const uint64_t ** boff, const uint32_t ** rlen, uint64_t * nreads)
{
	if ( !db || !bps || !bps_bytes || !boff || !rlen || !nreads ) return DACC_EINVAL;
	*bps = db->bps.data(); *bps_bytes = db->bps.size(); *boff = db->boff.data(); *rlen = db->rlen.data(); *nreads = db->rlen.size();
	return DACC_OK;
}

int dacc_db_write(const char * path, const uint8_t * bps, uint64_t bps_bytes, const uint64_t * boff, const uint32_t * rlen, uint64_t nreads)
{
	if ( !path || (nreads && (!bps || !boff || !rlen)) ) return DACC_EINVAL;
	std::vector<uint8_t> idx(IDX_HEADER + nreads*IDX_READ,0), out;
	int64_t totlen = 0; int32_t maxlen = 0;
	for ( uint64_t i = 0; i < nreads; ++i )
	{
		uint64_t const nb = (static_cast<uint64_t>(rlen[i])+3)/4;
		if ( boff[i] + nb > bps_bytes ) return DACC_EINVAL;
		uint8_t * r = idx.data() + IDX_HEADER + i*IDX_READ;
		put32(r+0,static_cast<int32_t>(i)); put32(r+4,rlen[i]); put32(r+8,0); put64(r+16,out.size()); put64(r+24,-1); put32(r+32,DB_BEST);
		out.insert(out.end(),bps+boff[i],bps+boff[i]+nb);
		totlen += rlen[i]; maxlen = std::max<int32_t>(maxlen,rlen[i]);
	}
	uint8_t * h = idx.data();
	put32(h+0,nreads); put32(h+4,nreads); put32(h+8,0); put32(h+12,1);       // ureads, treads, cutoff, all
	float const quarter = 0.25f; for ( int i = 0; i < 4; ++i ) std::memcpy(h+16+4*i,&quarter,4);
	put32(h+32,maxlen); put64(h+40,totlen); put32(h+48,nreads); put32(h+52,1);
	char stub[512];
	int const n = std::snprintf(stub,sizeof(stub),"files = %9d\n  %9d %s %s\nblocks = %9d\nsize = %10lld cutoff = %9d all = %1d\n %9d %9d\n %9d %9d\n",
		1,static_cast<int>(nreads),"synthetic","synthetic",1,200000000ll,0,1,0,0,static_cast<int>(nreads),static_cast<int>(nreads));
	if ( !writeFile(path,stub,n) || !writeFile(hidden(path,"idx"),idx.data(),idx.size()) || !writeFile(hidden(path,"bps"),out.data(),out.size()) ) return DACC_EINVAL;
	return DACC_OK;
}

int dacc_las_open(const char * path, dacc_las ** out)
{
	if ( !path || !out ) return DACC_EINVAL;
	dacc_las * las = new (std::nothrow) dacc_las; *out = las;
	if ( !las ) return DACC_ENOMEM;
	// The file is never held in memory (daccord.cpp:2133-2181 reads the byte range of a pile through the index): one
	// sequential pass over the record headers builds the A read -> byte offset table, unless a sidecar index from an
	// earlier run is there; dacc_las_piles then reads exactly the bytes of the A reads it is asked for.
	try
	{
		las->path = path;
		las->fd = ::open(path,O_RDONLY);
		if ( las->fd < 0 ) { las->err = std::string("cannot open ") + path; return DACC_EINVAL; }
		struct stat st;
		if ( ::fstat(las->fd,&st) != 0 || st.st_size < 12 ) { las->err = "overlap file too short"; return DACC_EINVAL; }
		las->fsize = static_cast<uint64_t>(st.st_size);
		uint8_t hdr[12];
		if ( !preadAll(las->fd,hdr,12,0) ) { las->err = "overlap file too short"; return DACC_EINVAL; }
		las->novl = get64(hdr); las->tspace = get32(hdr+8);
		if ( las->novl < 0 || las->tspace <= 0 || static_cast<uint64_t>(las->novl) > (las->fsize-12)/40 )
		{ las->err = "bad overlap file header"; return DACC_EINVAL; }
		las->tbytes = las->tspace <= 125 ? 1 : 2;                                  // TRACE_XOVR of align.h
		if ( sidecarEnabled() && loadSidecar(*las) ) { las->index_from_sidecar = true; return DACC_OK; }
		// scan: record headers through a 16 MB window, trace bytes are skipped (not copied anywhere)
		uint64_t const W = 1ull<<24;
		std::vector<uint8_t> win(W);
		uint64_t wlo = 0, whi = 0;           // file range held in win
		uint64_t pos = 12, lastrec = 0; int64_t prev = -1;
		std::vector<uint64_t> & aoff = las->aoff;
		for ( int64_t i = 0; i < las->novl; ++i )
		{
			if ( pos + 40 > las->fsize ) { las->err = "overlap file truncated"; return DACC_EINVAL; }
			if ( pos < wlo || pos + 40 > whi )
			{
				uint64_t const n = std::min<uint64_t>(W,las->fsize-pos);
				if ( !preadAll(las->fd,win.data(),n,pos) ) { las->err = "read error on the overlap file"; return DACC_EINVAL; }
				wlo = pos; whi = pos + n;
			}
			uint8_t const * r = win.data() + (pos-wlo);
			int32_t const tlen = get32(r+0), aread = get32(r+28), bread = get32(r+32);
			uint64_t const tb = static_cast<uint64_t>(tlen < 0 ? 0 : tlen)*las->tbytes;
			if ( tlen < 0 || pos + 40 + tb > las->fsize ) { las->err = "overlap file truncated (trace)"; return DACC_EINVAL; }
			if ( aread < 0 || bread < 0 ) { las->err = "negative read id in an overlap record"; return DACC_EINVAL; }
			if ( aread < prev ) { las->err = "records are not sorted by A read"; return DACC_EINVAL; }
			if ( aread != prev )
			{
				if ( prev < 0 ) las->minaread = aread;
				aoff.resize(static_cast<size_t>(aread)+1,pos);     // reads without records (prev+1..aread-1) share this offset
				prev = aread;
			}
			lastrec = pos; pos += 40 + tb;
		}
		if ( las->novl )
		{
			las->maxaread = prev;
			aoff.push_back(pos);
		}
		if ( sidecarEnabled() && las->novl ) writeSidecar(*las,lastrec);
	}
	catch ( std::bad_alloc const & ) { las->err = "out of memory"; return DACC_ENOMEM; }
	catch ( std::exception const & ex ) { las->err = ex.what(); return DACC_EINVAL; }
	return DACC_OK;
}
void dacc_las_close(dacc_las * las) { delete las; }
const char * dacc_las_error(dacc_las * las) { return las ? las->err.c_str() : "null handle"; }
int dacc_las_info(dacc_las * las, int64_t * novl, int32_t * tspace, int32_t * trace_bytes, int64_t * mina, int64_t * maxa)
{
	if ( !las ) return DACC_EINVAL;
	if ( novl ) *novl = las->novl; if ( tspace ) *tspace = las->tspace; if ( trace_bytes ) *trace_bytes = las->tbytes;
	if ( mina ) *mina = las->minaread; if ( maxa ) *maxa = las->maxaread;
	return DACC_OK;
}
int dacc_las_piles(dacc_las * las, int64_t afirst, int64_t alast, const dacc_pile ** piles, uint64_t * npiles,
	const dacc_overlap ** ovl, uint64_t * novl, const void ** trace, uint64_t * ntrace)
{
	if ( !las || !piles || !npiles || !ovl || !novl || !trace || !ntrace ) return DACC_EINVAL;
	try {
	las->opiles.clear(); las->oovl.clear(); las->otrace.clear();
	if ( afirst < 0 ) afirst = 0;
	if ( alast > las->maxaread+1 ) alast = las->maxaread+1;
	if ( afirst < alast )
	{
		uint64_t const lo = las->aoff[afirst], hi = las->aoff[alast];
		las->buf.resize(hi-lo);
		if ( hi > lo && !preadAll(las->fd,las->buf.data(),hi-lo,lo) ) { las->err = "read error on the overlap file"; return DACC_EINVAL; }
		uint64_t pos = 0; uint64_t const n = hi-lo;
		int64_t cur = -1;
		while ( pos < n )
		{
			if ( pos + 40 > n ) { las->err = "overlap file changed under the index (record header)"; return DACC_EINVAL; }
			uint8_t const * r = las->buf.data() + pos;
			dacc_overlap o; std::memset(&o,0,sizeof(o));
			o.tlen = get32(r+0); o.diffs = get32(r+4); o.abpos = get32(r+8); o.bbpos = get32(r+12); o.aepos = get32(r+16); o.bepos = get32(r+20);
			uint32_t fl; std::memcpy(&fl,r+24,4); o.flags = fl; o.aread = get32(r+28); o.bread = get32(r+32);
			uint64_t const tb = static_cast<uint64_t>(o.tlen < 0 ? 0 : o.tlen)*las->tbytes;
			if ( o.tlen < 0 || pos + 40 + tb > n || o.aread < afirst || o.aread >= alast || o.aread < cur || o.bread < 0 )
			{ las->err = "overlap file changed under the index (record)"; return DACC_EINVAL; }
			if ( o.aread != cur )
			{
				dacc_pile P; P.aread = o.aread; P.novl = 0; P.first_ovl = las->oovl.size();
				las->opiles.push_back(P); cur = o.aread;
			}
			o.trace_off = las->otrace.size()/las->tbytes;
			las->otrace.insert(las->otrace.end(),r+40,r+40+tb);
			las->oovl.push_back(o);
			las->opiles.back().novl += 1;
			pos += 40 + tb;
		}
	}
	*piles = las->opiles.data(); *npiles = las->opiles.size(); *ovl = las->oovl.data(); *novl = las->oovl.size();
	*trace = las->otrace.data(); *ntrace = las->otrace.size()/las->tbytes;
	} catch ( std::bad_alloc const & ) { las->err = "out of memory"; return DACC_ENOMEM; }
	return DACC_OK;
}

int dacc_las_write(const char * path, int32_t tspace, const dacc_overlap * ovl, uint64_t novl, const void * trace, uint64_t ntrace, int trace_bytes)
{
	if ( !path || tspace <= 0 || (novl && (!ovl || !trace)) ) return DACC_EINVAL;
	int const tb = tspace <= 125 ? 1 : 2;
	if ( trace_bytes != tb ) return DACC_EINVAL;
	std::vector<uint8_t> D(12);
	put64(D.data(),novl); put32(D.data()+8,tspace);
	uint8_t const * T = static_cast<uint8_t const *>(trace);
	for ( uint64_t i = 0; i < novl; ++i )
	{
		dacc_overlap const & o = ovl[i];
		if ( o.tlen < 0 || o.trace_off + o.tlen > ntrace ) return DACC_EINVAL;
		uint8_t r[40]; std::memset(r,0,sizeof(r));
		put32(r+0,o.tlen); put32(r+4,o.diffs); put32(r+8,o.abpos); put32(r+12,o.bbpos); put32(r+16,o.aepos); put32(r+20,o.bepos);
		uint32_t const fl = o.flags; std::memcpy(r+24,&fl,4); put32(r+28,o.aread); put32(r+32,o.bread);
		D.insert(D.end(),r,r+40);
		D.insert(D.end(),T + o.trace_off*tb,T + (o.trace_off+o.tlen)*tb);
	}
	return writeFile(path,D.data(),D.size()) ? DACC_OK : DACC_EINVAL;
}

}

// ---- read interval of a run (src/daccord.cpp:1115-1227) ----
#include <sstream>
namespace {
// "<int>,<int>" and nothing else (the reference parses with operator>>, get() == ',', operator>>, peek() == eof: :1122-1153)
bool parseIntPair(char const * text, int64_t & x, int64_t & y)
{
	std::istringstream is((std::string(text)));
	if ( !(is >> x) ) return false;
	int const c = is.get();
	if ( !is || c == std::istringstream::traits_type::eof() || c != ',' ) return false;
	if ( !(is >> y) ) return false;
	return is.peek() == std::istringstream::traits_type::eof();
}
void setErr(char * err, uint64_t const cap, std::string const & m) { if ( err && cap ) { std::snprintf(err,cap,"%s",m.c_str()); } }
}
extern "C" int dacc_read_interval(int64_t las_min, int64_t las_max, char const * J, char const * I, int64_t * minaread_out, int64_t * toparead_out, char * err, uint64_t errcap)
{
	if ( !minaread_out || !toparead_out ) return -1;
	int64_t minaread = las_min, maxaread = las_max;
	if ( J )
	{
		int64_t cnt, div;
		if ( !parseIntPair(J,cnt,div) ) { setErr(err,errcap,std::string("unable to parse ") + J); return -1; }
		int64_t const toparead = maxaread + 1, span = toparead > minaread ? toparead-minaread : 0;
		if ( span && !div ) { setErr(err,errcap,"denominator of J argument cannot be zero"); return -1; }
		if ( toparead > minaread )
		{
			int64_t const part = div ? (span + div - 1)/div : 0;
			int64_t const ilow = std::min(minaread + cnt*part,toparead), ihigh = std::min(ilow+part,toparead);
			if ( ihigh > ilow ) { minaread = ilow; maxaread = ihigh-1; } else { minaread = 0; maxaread = -1; }
		}
	}
	else if ( I )
	{
		int64_t lo, hi;
		if ( !parseIntPair(I,lo,hi) ) { setErr(err,errcap,std::string("unable to parse ") + I); return -1; }
		minaread = std::max(lo,minaread); maxaread = std::min(hi,maxaread);
	}
	*minaread_out = minaread;
	*toparead_out = maxaread >= 0 ? maxaread+1 : maxaread;
	return 0;
}
