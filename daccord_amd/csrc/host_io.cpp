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
	std::string err;
	std::vector<uint8_t> D;
	int64_t novl; int32_t tspace; int32_t tbytes;
	std::vector<dacc_overlap> ovl;          // all records, .las order
	std::vector<uint8_t> trace;             // all trace values, tbytes each
	std::vector<uint64_t> afirst;           // index of the first record of aread (size = maxaread+2)
	int64_t minaread, maxaread;
	std::vector<dacc_pile> opiles; std::vector<dacc_overlap> oovl; std::vector<uint8_t> otrace;
};

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
	las->novl = 0; las->tspace = 0; las->tbytes = 1; las->minaread = 0; las->maxaread = -1;
	// one pass over the file with a bounded buffer: the records go straight into the overlap / trace arrays (the file
	// itself is never held in memory); no exception leaves this function
	try
	{
		FILE * f = std::fopen(path,"rb");
		if ( !f ) { las->err = std::string("cannot open ") + path; return DACC_EINVAL; }
		struct Closer { FILE * f; ~Closer() { std::fclose(f); } } closer = { f };
		std::vector<char> iobuf(1u<<22); std::setvbuf(f,iobuf.data(),_IOFBF,iobuf.size());
		uint8_t hdr[12];
		if ( std::fread(hdr,1,12,f) != 12 ) { las->err = "overlap file too short"; return DACC_EINVAL; }
		las->novl = get64(hdr); las->tspace = get32(hdr+8);
		std::fseek(f,0,SEEK_END); long const fsize = std::ftell(f); std::fseek(f,12,SEEK_SET);
		if ( las->novl < 0 || las->tspace <= 0 || fsize < 12 || static_cast<uint64_t>(las->novl) > static_cast<uint64_t>(fsize-12)/40 )
		{ las->err = "bad overlap file header"; return DACC_EINVAL; }
		las->tbytes = las->tspace <= 125 ? 1 : 2;                                  // TRACE_XOVR of align.h
		las->ovl.reserve(las->novl);
		las->trace.reserve(static_cast<size_t>(fsize-12) - static_cast<size_t>(las->novl)*40);
		int64_t prev = -1;
		uint64_t left = static_cast<uint64_t>(fsize-12);
		for ( int64_t i = 0; i < las->novl; ++i )
		{
			uint8_t r[40];
			if ( left < 40 || std::fread(r,1,40,f) != 40 ) { las->err = "overlap file truncated"; return DACC_EINVAL; }
			left -= 40;
			dacc_overlap o; std::memset(&o,0,sizeof(o));
			o.tlen = get32(r+0); o.diffs = get32(r+4); o.abpos = get32(r+8); o.bbpos = get32(r+12); o.aepos = get32(r+16); o.bepos = get32(r+20);
			uint32_t fl; std::memcpy(&fl,r+24,4); o.flags = fl; o.aread = get32(r+28); o.bread = get32(r+32);
			size_t const tb = static_cast<size_t>(o.tlen < 0 ? 0 : o.tlen)*las->tbytes;
			if ( o.tlen < 0 || tb > left ) { las->err = "overlap file truncated (trace)"; return DACC_EINVAL; }
			if ( o.aread < 0 || o.bread < 0 ) { las->err = "negative read id in an overlap record"; return DACC_EINVAL; }
			o.trace_off = las->trace.size()/las->tbytes;
			size_t const t0 = las->trace.size(); las->trace.resize(t0+tb);
			if ( tb && std::fread(las->trace.data()+t0,1,tb,f) != tb ) { las->err = "overlap file truncated (trace)"; return DACC_EINVAL; }
			left -= tb;
			if ( o.aread < prev ) { las->err = "records are not sorted by A read"; return DACC_EINVAL; }
			prev = o.aread;
			las->ovl.push_back(o);
		}
		if ( las->novl )
		{
			las->minaread = las->ovl.front().aread; las->maxaread = las->ovl.back().aread;
			las->afirst.assign(las->maxaread+2,las->ovl.size());
			for ( size_t i = las->ovl.size(); i-- > 0; ) las->afirst[las->ovl[i].aread] = i;
			for ( int64_t a = las->maxaread; a >= 0; --a ) if ( las->afirst[a] > las->afirst[a+1] ) las->afirst[a] = las->afirst[a+1];
		}
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
	for ( int64_t a = afirst; a < alast; ++a )
	{
		uint64_t const lo = las->afirst[a], hi = las->afirst[a+1];
		if ( lo == hi ) continue;
		dacc_pile P; P.aread = a; P.novl = hi-lo; P.first_ovl = las->oovl.size();
		for ( uint64_t i = lo; i < hi; ++i )
		{
			dacc_overlap o = las->ovl[i];
			size_t const tb = static_cast<size_t>(o.tlen)*las->tbytes;
			uint8_t const * src = las->trace.data() + o.trace_off*las->tbytes;
			o.trace_off = las->otrace.size()/las->tbytes;
			las->otrace.insert(las->otrace.end(),src,src+tb);
			las->oovl.push_back(o);
		}
		las->opiles.push_back(P);
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
