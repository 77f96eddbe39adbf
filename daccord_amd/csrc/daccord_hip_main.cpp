/*
 * daccord_hip -- the `daccord` command line on the MI355X path:
 *
 *     daccord_hip [options] reads.las reads.db [reads2.db]
 *
 * Host program above the C ABI (include/daccord_hip.h, include/daccord_io.h); the argument grammar, the read interval
 * logic and the ordered FASTA output follow src/daccord.cpp:
 *   :185-207, 1282-1305  options (flag and value joined: -w40 -k8 -I0,99; options precede the positionals)
 *   :1156-1183           -J<i,j>  part i of j of the A-read range (takes precedence over -I)
 *   :1222-1230           -I<lo,hi> both ends inclusive, intersected with the range of the .las
 *   :2120-2126           --vard<v> per-read depth cap instead of -D
 *   :2107-2112, 2481-2534 piles in A-read order, output in A-read order
 *   HandleContext.hpp:2710-2724 FASTA records; the middle name field is numbered sequentially (the reference's -t1
 *                        numbering; with -t>1 its numbering depends on the thread schedule)
 *   :2464-2478           a read that fails is logged on stderr and skipped
 * -t and -T are accepted and ignored (no host worker threads, no temporary files).  The error profile comes from
 * --eprof<p_i,p_d,est_cor>, from -E<file> / <las>.eprof -- the three numbers as text (our own form) or the reference's 48-byte
 * binary form (src/daccord.cpp:1855-1864: libmaus2 AlignmentStatistics + two doubles; the form is detected) --, or is estimated from the
 * first 1024 piles like src/daccord.cpp:1653-1878 does and written to that file as text, with --binaryeprof in the reference's form
 * (--eprofonly: stop there; no GPU needed).
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>
#include <algorithm>
#include <fstream>
#include <sstream>
#include <iostream>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <deque>
#include <chrono>
#include <sys/stat.h>
#include <unistd.h>
#include "../../include/daccord_hip.h"
#include "../../include/daccord_io.h"

namespace {

struct Options
{
	uint32_t w = 40, a = 10, m = 3; uint64_t d = UINT64_MAX, e = UINT64_MAX, l = 0, D = 5000, vard = 0;
	bool f = false; int V = 1; bool haveI = false, haveJ = false; std::string Itext, Jtext; int64_t Ilo = 0, Ihi = 0, Jc = 0, Jd = 1;
	std::string E, eprof; uint32_t klow = 8, khigh = 8; int32_t minff = 0, maxff = 2;
	bool eprofonly = false, keepeprof = false, deepprofileonly = false, binaryeprof = false, loaderonly = false; int device = 0; int gpus = 1; uint64_t batch = 2000;
	std::vector<std::string> pos;
};

[[noreturn]] void die(std::string const & m) { std::fprintf(stderr,"[E] %s\n",m.c_str()); std::exit(EXIT_FAILURE); }

bool parsePair(std::string const & s, int64_t & x, int64_t & y)
{
	std::istringstream is(s); char c = 0;
	if ( !(is >> x) ) return false;
	if ( !(is.get(c)) || c != ',' ) return false;
	if ( !(is >> y) ) return false;
	return is.peek() == std::istringstream::traits_type::eof();
}

uint64_t num(std::string const & opt, std::string const & v)
{
	if ( v.empty() ) die("option " + opt + " needs a value");
	char * end = 0; unsigned long long const x = std::strtoull(v.c_str(),&end,10);
	if ( *end ) die("unable to parse " + opt + v);
	return x;
}

Options parse(int argc, char ** argv)
{
	Options o;
	for ( int i = 1; i < argc; ++i )
	{
		std::string const a = argv[i];
		if ( !o.pos.empty() || a.size() < 2 || a[0] != '-' ) { o.pos.push_back(a); continue; }
		if ( a[1] == '-' )
		{
			auto val = [&](char const * name, std::string & out) -> bool {
				std::string const p = std::string("--") + name;
				if ( a.compare(0,p.size(),p) != 0 ) return false;
				out = a.substr(p.size()); if ( !out.empty() && out[0] == '=' ) out = out.substr(1);
				return true;
			};
			std::string v;
			if ( val("minfilterfreq",v) ) o.minff = static_cast<int32_t>(num("--minfilterfreq",v));
			else if ( val("maxfilterfreq",v) ) o.maxff = static_cast<int32_t>(num("--maxfilterfreq",v));
			else if ( val("vard",v) ) o.vard = num("--vard",v);
			else if ( val("eprofonly",v) ) o.eprofonly = v.empty() || v != "0";
			else if ( val("keepeprof",v) ) o.keepeprof = v.empty() || v != "0";
			else if ( val("binaryeprof",v) ) o.binaryeprof = v.empty() || v != "0";
			else if ( val("eprof",v) ) o.eprof = v;
			else if ( val("device",v) ) o.device = static_cast<int>(num("--device",v));
			else if ( val("gpus",v) ) { o.gpus = static_cast<int>(num("--gpus",v)); if ( o.gpus < 1 || o.gpus > 64 ) die("--gpus needs a number between 1 and 64"); }
			else if ( val("batch",v) ) { o.batch = num("--batch",v); if ( !o.batch ) die("--batch needs a positive number of A reads"); }
			else if ( val("deepprofileonly",v) ) o.deepprofileonly = v.empty() || v != "0";
			else if ( val("loaderonly",v) ) o.loaderonly = v.empty() || v != "0";
			else die("unknown option " + a);
			continue;
		}
		char const key = a[1]; std::string const v = a.substr(2);
		switch ( key )
		{
			case 'w': o.w = num("-w",v); break;
			case 'a': o.a = num("-a",v); break;
			case 'm': o.m = num("-m",v); break;
			case 'l': o.l = num("-l",v); break;
			case 'D': o.D = num("-D",v); break;
			case 'd': o.d = num("-d",v); break;
			case 'e': o.e = num("-e",v); break;
			case 'V': o.V = v.empty() ? 1 : static_cast<int>(num("-V",v)); break;
			case 'f': o.f = v.empty() ? true : (num("-f",v) != 0); break;
			case 't': case 'T': break;
			case 'E': o.E = v; break;
			case 'I': o.haveI = true; o.Itext = v; if ( !parsePair(v,o.Ilo,o.Ihi) ) die("unable to parse " + v); break;
			case 'J': o.haveJ = true; o.Jtext = v; if ( !parsePair(v,o.Jc,o.Jd) ) die("unable to parse " + v); break;
			case 'k':
			{
				int64_t x, y;
				if ( parsePair(v,x,y) ) { o.klow = x; o.khigh = y; } else { o.klow = o.khigh = num("-k",v); }
				break;
			}
			default: die("unknown option " + a);
		}
	}
	if ( o.pos.size() < 2 )
	{
		std::fprintf(stderr,"usage: daccord_hip [options] reads.las reads.db [reads2.db]\n"
			"  -w<40> window  -a<10> advance  -k<8|lo,hi> k-mer size  -d<max depth>  -D<5000> max alignments per read  --vard<v>\n"
			"  -m<3> min window coverage  -e<max window error>  -l<0> min output length  -f produce full reads\n"
			"  -I<lo,hi> read interval (inclusive)  -J<i,j> part i of j  --minfilterfreq<0> --maxfilterfreq<2>\n"
			"  --eprof<p_i,p_d,est_cor> | -E<file> error profile (default: <las>.eprof, estimated if missing)  --eprofonly  --keepeprof  --deepprofileonly\n"
			"  --binaryeprof             write an estimated profile in the reference's binary form (read in either form)\n"
			"  --device<0> first HIP device  --gpus<1> devices used by this process (batches are dealt to them, output stays ordered)\n"
			"  --batch<2000> A reads per GPU batch  (or one process per GPU like the reference: -J<g,G> --device<g>)\n"
			"  --loaderonly  measure the host side alone (loader + selection + planner, --gpus planner threads), no device\n");
		std::exit(EXIT_FAILURE);
	}
	// what dacc_create would refuse, said before any file is read and in the option's own words.  The reference compiles k = 3 ... 12 and
	// throws "k-mer size k is not compiled in" beyond (DebruijnGraphContainer.hpp:41-110); its k-mer words hold 32 bits (DebruijnGraph.hpp:956-961),
	// which defines the algorithm up to k = 16 -- what this build runs; k = 17 has no reference semantics.
	if ( o.klow < 3 || o.khigh > 16 || o.klow > o.khigh )
		die("k-mer size " + std::to_string(o.klow) + (o.khigh != o.klow ? "," + std::to_string(o.khigh) : std::string()) + " is not compiled in (3 <= k <= 16, low <= high)");
	if ( !o.w || o.w > 128 ) die("-w must be in [1,128]");      // DACC_WMAX (dev_types.hpp): two 64 bit words of the consensus -> A alignment
	if ( !o.a ) die("-a must be at least 1");
	if ( o.minff < 0 || o.maxff < o.minff ) die("--minfilterfreq / --maxfilterfreq: need 0 <= min <= max");
	return o;
}

// The reference's own .eprof (daccord.cpp:1855-1860: GAS.serialise + two serialiseDouble; read back by GAS.deserialise, :1864): 48
// bytes -- libmaus2 AlignmentStatistics {matches, mismatches, insertions, deletions} as four 8-byte big-endian numbers
// (NumberSerialisation::serialiseNumber), then eavg and edif as raw host-order doubles.  libmaus2 is not in the reference tree: the layout
// is its published one, restated here, and has not met a file written by the reference.  The rates follow from the counts
// exactly as daccord.cpp:1867-1878 computes them after deserialising.
size_t const refProfileBytes = 48;
bool decodeProfileBinary(std::string const & s, double v[3], uint64_t counts[4])
{
	if ( s.size() != refProfileBytes ) return false;
	for ( int i = 0; i < 4; ++i )
	{
		uint64_t x = 0;
		for ( int j = 0; j < 8; ++j ) x = (x << 8) | static_cast<uint8_t>(s[8*i+j]);
		counts[i] = x;
	}
	// matches + mismatches + deletions is a sum of base counts of at most 1024 piles: anything near 2^63 is not a profile
	for ( int i = 0; i < 4; ++i ) if ( counts[i] >> 56 ) return false;
	uint64_t const len = counts[0] + counts[1] + counts[3], numerr = counts[1] + counts[3] + counts[2];
	if ( !len ) return false;
	v[0] = static_cast<double>(counts[2])/len; v[1] = static_cast<double>(counts[3])/len; v[2] = 1.0 - static_cast<double>(numerr)/len;
	return true;
}
std::string encodeProfileBinary(uint64_t const counts[4], double eavg, double edif)
{
	std::string s(refProfileBytes,'\0');
	for ( int i = 0; i < 4; ++i ) for ( int j = 0; j < 8; ++j ) s[8*i+j] = static_cast<char>((counts[i] >> (8*(7-j))) & 0xff);
	std::memcpy(&s[32],&eavg,8); std::memcpy(&s[40],&edif,8);
	return s;
}

// an error profile file: our text form (three numbers p_i p_d est_cor, blanks or commas between them) or the reference's binary form;
// *binary says which one it was
bool readProfile(std::string const & fn, double v[3], bool * binary)
{
	std::ifstream in(fn.c_str(),std::ios::binary);
	if ( !in ) return false;
	std::string s((std::istreambuf_iterator<char>(in)),std::istreambuf_iterator<char>());
	bool text = !s.empty();
	for ( size_t i = 0; i < s.size() && text; ++i )
	{
		unsigned char const c = static_cast<unsigned char>(s[i]);
		text = (c >= 0x20 && c < 0x7f) || c == '\n' || c == '\r' || c == '\t';
	}
	if ( binary ) *binary = !text;
	if ( !text ) { uint64_t counts[4]; return decodeProfileBinary(s,v,counts); }
	for ( size_t i = 0; i < s.size(); ++i ) if ( s[i] == ',' ) s[i] = ' ';
	std::istringstream is(s);
	return static_cast<bool>(is >> v[0] >> v[1] >> v[2]);
}

struct Db { dacc_db * h = 0; uint8_t const * bps = 0; uint64_t nbytes = 0; uint64_t const * boff = 0; uint32_t const * rlen = 0; uint64_t n = 0; };
void openDb(std::string const & fn, Db & d)
{
	if ( dacc_db_open(fn.c_str(),&d.h) ) die("cannot open database " + fn + ": " + (d.h ? dacc_db_error(d.h) : "out of memory"));
	if ( dacc_db_arrays(d.h,&d.bps,&d.nbytes,&d.boff,&d.rlen,&d.n) ) die(std::string("database: ") + dacc_db_error(d.h));
}

}

int main(int argc, char ** argv)
{
	Options const o = parse(argc,argv);
	std::string const lasfn = o.pos[0];
	dacc_las * las = 0;
	if ( dacc_las_open(lasfn.c_str(),&las) ) die("cannot open " + lasfn + ": " + (las ? dacc_las_error(las) : "out of memory"));
	int64_t novl = 0, lasmin = 0, lasmax = -1; int32_t tspace = 0, tbytes = 1;
	if ( dacc_las_info(las,&novl,&tspace,&tbytes,&lasmin,&lasmax) ) die(std::string("las: ") + dacc_las_error(las));

	Db A, B2; openDb(o.pos[1],A);
	bool const twodb = o.pos.size() > 2;
	if ( twodb && o.vard ) die("vard option is not supported for asymmetric (DB1 != DB2) alignments");       // daccord.cpp:1342-1348
	std::vector<uint8_t> bps; std::vector<uint64_t> boff; std::vector<uint32_t> rlen;
	uint64_t nA = 0;
	if ( twodb )
	{
		// asymmetric mode (daccord.cpp:1337-1364): A reads from the first database, B reads from the second; on the device
		// both live in one read store, ids of B shifted behind A's
		openDb(o.pos[2],B2);
		bps.assign(A.bps,A.bps+A.nbytes); bps.insert(bps.end(),B2.bps,B2.bps+B2.nbytes);
		boff.assign(A.boff,A.boff+A.n); for ( uint64_t i = 0; i < B2.n; ++i ) boff.push_back(B2.boff[i]+A.nbytes);
		rlen.assign(A.rlen,A.rlen+A.n); rlen.insert(rlen.end(),B2.rlen,B2.rlen+B2.n);
		nA = A.n;
	}
	uint8_t const * const pbps = twodb ? bps.data() : A.bps; uint64_t const nbps = twodb ? bps.size() : A.nbytes;
	uint64_t const * const pboff = twodb ? boff.data() : A.boff; uint32_t const * const prlen = twodb ? rlen.data() : A.rlen;
	uint64_t const nreads = twodb ? rlen.size() : A.n;
	// average read length of the B database (DB2.getAverageReadLength(), daccord.cpp:1366): --vard scales with it
	Db const & DB2 = twodb ? B2 : A;
	uint64_t tot = 0; for ( uint64_t i = 0; i < DB2.n; ++i ) tot += DB2.rlen[i];
	uint64_t const avgrl = DB2.n ? tot/DB2.n : 0;

	// read interval (daccord.cpp:1115-1227): dacc_read_interval, which tests/test_oracle_vs_ref.py runs against the reference's lines
	int64_t minaread = 0, toparead = 0;
	{
		char ebuf[256]; ebuf[0] = 0;
		if ( dacc_read_interval(lasmin,lasmax,o.haveJ ? o.Jtext.c_str() : 0,o.haveI ? o.Itext.c_str() : 0,&minaread,&toparead,ebuf,sizeof(ebuf)) ) die(ebuf);
	}
	int64_t const maxaread = toparead - 1;     // (-2 for an empty interval: only compared against minaread below)
	if ( o.V ) std::fprintf(stderr,"[V] minaread=%lld toparead=%lld\n",static_cast<long long>(minaread),static_cast<long long>(toparead));

	// the .las must belong to this database: the reference's readers throw on an id beyond it
	if ( novl && lasmax >= static_cast<int64_t>(A.n) )
		die("A read id " + std::to_string(lasmax) + " of " + lasfn + " is beyond the " + std::to_string(A.n) + " reads of " + o.pos[1] + ": the .las does not belong to this database");
	if ( o.vard && !avgrl ) die("--vard needs a database with reads (average read length is zero)");
	uint64_t const nB = twodb ? B2.n : A.n;

	// batch of piles [b0,b1): load, top-D select per pile, shift B ids in two database mode
	struct Batch { std::vector<dacc_overlap> sel; std::vector<dacc_pile> spiles; std::vector<uint8_t> trace; uint64_t ntrace = 0; bool end = false; std::string err; uint64_t seq = 0; };
	std::vector<dacc_overlap> tmp;
	auto loadBatch = [&](int64_t const b0, int64_t const b1, bool const lowest, Batch & B) -> void
	{
		dacc_pile const * piles = 0; uint64_t npiles = 0; dacc_overlap const * ovl = 0; uint64_t no = 0; void const * trace = 0; uint64_t ntrace = 0;
		B.sel.clear(); B.spiles.clear(); B.trace.clear(); B.ntrace = 0; B.err.clear();
		if ( dacc_las_piles(las,b0,b1,&piles,&npiles,&ovl,&no,&trace,&ntrace) ) { B.err = std::string("las: ") + dacc_las_error(las); return; }
		for ( uint64_t z = 0; z < no; ++z )
			if ( static_cast<uint64_t>(ovl[z].bread) >= nB )
			{ B.err = "B read id " + std::to_string(ovl[z].bread) + " (A read " + std::to_string(ovl[z].aread) + ") is beyond the " + std::to_string(nB) + " reads of the database"; return; }
		for ( uint64_t i = 0; i < npiles; ++i )
		{
			uint64_t const rl = prlen[piles[i].aread];
			uint64_t const lmaxinput = o.vard ? std::max<uint64_t>(std::max<uint64_t>(2*o.vard,1),
				static_cast<uint64_t>((2.0*static_cast<double>(o.vard)*static_cast<double>(rl)/static_cast<double>(avgrl))+0.5)) : o.D;
			tmp.resize(std::max<uint64_t>(piles[i].novl,1)); uint64_t nout = 0;
			int const rc = lowest ? dacc_pile_select_lowest(ovl+piles[i].first_ovl,piles[i].novl,tbytes,lmaxinput,tmp.data(),&nout)
			                      : dacc_pile_select(ovl+piles[i].first_ovl,piles[i].novl,tbytes,lmaxinput,tmp.data(),&nout);
			if ( rc ) { B.err = "pile selection failed"; return; }
			dacc_pile q; q.aread = piles[i].aread; q.novl = nout; q.first_ovl = B.sel.size();
			for ( uint64_t z = 0; z < nout; ++z ) { dacc_overlap v = tmp[z]; v.bread += nA; B.sel.push_back(v); }
			B.spiles.push_back(q);
		}
		// the handle's arrays are valid until its next call, which the loader makes while this batch is on the GPU
		B.trace.assign(static_cast<uint8_t const *>(trace),static_cast<uint8_t const *>(trace) + ntrace*static_cast<uint64_t>(tbytes));
		B.ntrace = ntrace;
	};

	// error profile: --eprof, else the file; estimated if the file is missing or older than the .las (unless --keepeprof),
	// daccord.cpp:1653-1660
	double prof[3] = {0,0,0};
	std::string const eproffn = o.E.empty() ? (lasfn + ".eprof") : o.E;
	bool have = false;
	if ( !o.eprof.empty() )
	{
		std::string s = o.eprof; for ( size_t i = 0; i < s.size(); ++i ) if ( s[i] == ',' ) s[i] = ' ';
		std::istringstream is(s); if ( !(is >> prof[0] >> prof[1] >> prof[2]) ) die("--eprof needs three numbers: p_i,p_d,est_cor");
		have = true;
	}
	else
	{
		struct stat se, sl;
		bool const exists = ::stat(eproffn.c_str(),&se) == 0;
		bool const older = exists && ::stat(lasfn.c_str(),&sl) == 0 &&
			( se.st_mtim.tv_sec < sl.st_mtim.tv_sec || (se.st_mtim.tv_sec == sl.st_mtim.tv_sec && se.st_mtim.tv_nsec < sl.st_mtim.tv_nsec) );
		if ( exists && !(older && !o.keepeprof) )
		{
			bool bin = false;
			if ( !readProfile(eproffn,prof,&bin) ) die("cannot parse the error profile " + eproffn + " (three numbers p_i p_d est_cor as text, or the reference's 48-byte binary profile)");
			if ( o.V ) std::fprintf(stderr,"[V] error profile %s read (%s)\n",eproffn.c_str(),bin ? "reference binary form" : "text");
			have = true;
		}
	}
	if ( o.deepprofileonly )
	{
		// --deepprofileonly (daccord.cpp:1442-1650): distribution of the window error rates of the estimator's windows over the
		// first 4096 piles of the interval, "[deep] <error rate> <fraction of the windows at or below it>" per distinct value
		int64_t const top = std::min<int64_t>(toparead,minaread+4096);
		dacc_eprof * ep = 0;
		if ( dacc_eprof_create(&ep,tspace,pbps,pboff,prlen,nreads,twodb ? 1 : 0) ) die("error distribution estimate: out of memory");
		dacc_eprof_set_deep(ep,1);
		unsigned int hw = std::thread::hardware_concurrency(); if ( !hw ) hw = 1; if ( hw > 64 ) hw = 64;
		Batch EB;
		for ( int64_t b0 = minaread; b0 < top; b0 += 256 )
		{
			loadBatch(b0,std::min<int64_t>(top,b0+256),true,EB);
			if ( !EB.err.empty() ) die(EB.err);
			if ( EB.spiles.empty() ) continue;
			if ( dacc_eprof_add(ep,EB.spiles.data(),EB.spiles.size(),EB.sel.data(),EB.sel.size(),EB.trace.data(),EB.ntrace,tbytes,o.d,hw) ) die("error distribution estimate failed (out of memory)");
		}
		uint32_t const * dv = 0; uint64_t dn = 0;
		dacc_eprof_deep(ep,&dv,&dn);
		double const dcnt = static_cast<double>(dn);
		uint64_t i = 0;
		while ( i < dn )
		{
			uint64_t j = i; while ( j < dn && dv[j] == dv[i] ) ++j;
			std::printf("[deep]\t%g\t%g\n",dv[i]/4294967295.0,static_cast<double>(j)/dcnt);
			i = j;
		}
		dacc_eprof_destroy(ep);
		dacc_las_close(las); dacc_db_close(A.h); if ( twodb ) dacc_db_close(B2.h);
		return EXIT_SUCCESS;
	}
	if ( !have && !(minaread <= maxaread) )
	{
		// an empty part (-J beyond the data) has nothing to estimate from and nothing to correct: the reference emits nothing
		if ( o.V ) std::fprintf(stderr,"[V] empty read interval, nothing to do\n");
		dacc_las_close(las); dacc_db_close(A.h); if ( twodb ) dacc_db_close(B2.h);
		return EXIT_SUCCESS;
	}
	if ( !have )
	{
		// estimate from the first 1024 piles of the interval (daccord.cpp:1653-1878): k=8, w=40, a=5
		auto const te0 = std::chrono::steady_clock::now();
		int64_t const top = std::min<int64_t>(toparead,minaread+1024);
		uint64_t counts[4] = {0,0,0,0}; uint64_t usable = 0, unusable = 0; double eavg = 0, edif = 0;
		dacc_eprof * ep = 0;
		if ( dacc_eprof_create(&ep,tspace,pbps,pboff,prlen,nreads,twodb ? 1 : 0) ) die("error profile estimation: out of memory");
		unsigned int hw = std::thread::hardware_concurrency(); if ( !hw ) hw = 1; if ( hw > 64 ) hw = 64;
		Batch EB;
		for ( int64_t b0 = minaread; b0 < top; b0 += 256 )
		{
			loadBatch(b0,std::min<int64_t>(top,b0+256),true,EB);
			if ( !EB.err.empty() ) die(EB.err);
			if ( EB.spiles.empty() ) continue;
			if ( dacc_eprof_add(ep,EB.spiles.data(),EB.spiles.size(),EB.sel.data(),EB.sel.size(),EB.trace.data(),EB.ntrace,tbytes,o.d,hw) ) die("error profile estimation failed (out of memory)");
		}
		{
			// malformed piles are left out of the estimate (one read is logged and skipped, daccord.cpp:2464-2478); say so, and refuse
			// a profile made from a minority of the sample (a .las that does not belong to this database)
			uint64_t nskip = 0, nseen = 0;
			dacc_eprof_skipped(ep,&nskip,&nseen);
			if ( nskip ) std::fprintf(stderr,"[E] error profile estimation: %llu of %llu piles left out (malformed overlap or trace records)\n",static_cast<unsigned long long>(nskip),static_cast<unsigned long long>(nseen));
			if ( nseen && 2*nskip > nseen ) die("error profile estimation: most piles of the sample have malformed overlap records -- does the .las belong to this database?");
		}
		if ( dacc_eprof_finish(ep,counts,&usable,&unusable,&eavg,&edif,prof) ) die("error profile estimation found no usable window; give --eprof<p_i,p_d,est_cor>");
		dacc_eprof_destroy(ep);
		std::fprintf(stderr,"usable=%llu unusable=%llu eavg=%g edif=%g\n",static_cast<unsigned long long>(usable),static_cast<unsigned long long>(unusable),eavg,edif);
		std::fprintf(stderr,"AlignmentStatistics(matches=%llu,mismatches=%llu,insertions=%llu,deletions=%llu)\n",
			static_cast<unsigned long long>(counts[0]),static_cast<unsigned long long>(counts[1]),static_cast<unsigned long long>(counts[2]),static_cast<unsigned long long>(counts[3]));
		if ( o.V ) std::fprintf(stderr,"[V] error profile estimated on %u host threads in %.2f s\n",hw,std::chrono::duration<double>(std::chrono::steady_clock::now()-te0).count());
		// temp file + rename (daccord.cpp:1855-1860): parallel -J jobs on one .las never see a partly written profile
		std::string const tmpfn = eproffn + ".tmp." + std::to_string(static_cast<long long>(::getpid()));
		{
			std::ofstream out(tmpfn.c_str(),std::ios::binary);
			if ( o.binaryeprof ) out << encodeProfileBinary(counts,eavg,edif);       // what the reference's GAS.deserialise reads (daccord.cpp:1864)
			else { char buf[128]; std::snprintf(buf,sizeof(buf),"%.17g %.17g %.17g\n",prof[0],prof[1],prof[2]); out << buf; }
			out.flush();
			if ( !out ) { std::remove(tmpfn.c_str()); die("cannot write the error profile " + tmpfn); }
		}
		if ( std::rename(tmpfn.c_str(),eproffn.c_str()) != 0 ) { std::remove(tmpfn.c_str()); die("cannot rename " + tmpfn + " to " + eproffn); }
	}
	if ( o.V ) std::fprintf(stderr,"[V] p_i=%.17g p_d=%.17g est_cor=%.17g\n",prof[0],prof[1],prof[2]);
	if ( o.eprofonly ) return EXIT_SUCCESS;

	dacc_params p; std::memset(&p,0,sizeof(p));
	p.w = o.w; p.a = o.a; p.klow = o.klow; p.khigh = o.khigh; p.minfilterfreq = o.minff; p.maxfilterfreq = o.maxff; p.minwindowcov = o.m;
	p.maxalign = o.d; p.eminrate = o.e; p.minlen = o.l; p.producefull = o.f ? 1 : 0; p.tspace = tspace; p.device = o.device; p.verbose = o.V;
	if ( o.loaderonly )
	{
		// Measurement without a device (round 5, VERDICT r04 task 5): the host side of a run alone.  One loader thread reads and selects
		// the batches exactly as below; --gpus N consumer threads stand in for the device workers and run only what dacc_submit_piles
		// does on the host before its first upload (the planner, dacc_plan_only).  Reports what the loader sustains (piles/s, A-read
		// Mbase/s) and what the planners do, so that "can one loader feed N GPUs" has a number.
		int const nwork = o.gpus;
		int64_t const batch = static_cast<int64_t>(o.batch);
		std::mutex mu; std::condition_variable cv; std::deque<Batch *> loaded, freeb; std::vector<Batch> bslots(2*nwork);
		for ( size_t i = 0; i < bslots.size(); ++i ) freeb.push_back(&bslots[i]);
		double load_s = 0, wait_s = 0, plan_s = 0; uint64_t npl = 0, nov = 0, abases = 0, nwin = 0, nbat = 0; std::string fatal;
		auto const t0 = std::chrono::steady_clock::now();
		std::thread loader([&]() {
			uint64_t seq = 0;
			for ( int64_t b0 = minaread; ; b0 += batch )
			{
				Batch * B;
				auto const tw = std::chrono::steady_clock::now();
				{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk,[&]{ return !freeb.empty(); }); B = freeb.front(); freeb.pop_front(); }
				auto const tl = std::chrono::steady_clock::now();
				B->end = !(b0 < toparead); B->seq = seq++;
				if ( !B->end ) loadBatch(b0,std::min(toparead,b0+batch),false,*B);
				auto const te = std::chrono::steady_clock::now();
				bool const stop = B->end || !B->err.empty();
				{
					std::lock_guard<std::mutex> lk(mu);
					wait_s += std::chrono::duration<double>(tl-tw).count(); load_s += std::chrono::duration<double>(te-tl).count();
					if ( !B->end ) { npl += B->spiles.size(); nov += B->sel.size(); for ( size_t i = 0; i < B->spiles.size(); ++i ) abases += prlen[B->spiles[i].aread]; }
					loaded.push_back(B);
				}
				cv.notify_all();
				if ( stop ) break;
			}
		});
		bool stopall = false;
		auto const consumer = [&]()
		{
			while ( true )
			{
				Batch * B = 0;
				{
					std::unique_lock<std::mutex> lk(mu);
					cv.wait(lk,[&]{ return stopall || !loaded.empty(); });
					if ( stopall && loaded.empty() ) return;
					B = loaded.front(); loaded.pop_front();
					if ( B->end || !B->err.empty() ) { stopall = true; if ( !B->err.empty() && fatal.empty() ) fatal = B->err; }
				}
				cv.notify_all();
				if ( !B->end && B->err.empty() && !B->spiles.empty() )
				{
					auto const tp = std::chrono::steady_clock::now();
					uint64_t w = 0, nb = 0;
					int const rc = dacc_plan_only(&p,prlen,nreads,B->spiles.data(),B->spiles.size(),B->sel.data(),B->sel.size(),B->trace.data(),B->ntrace,tbytes,&w,&nb);
					double const dt = std::chrono::duration<double>(std::chrono::steady_clock::now()-tp).count();
					std::lock_guard<std::mutex> lk(mu);
					if ( rc && fatal.empty() ) fatal = "planner failed (" + std::to_string(rc) + ")";
					plan_s += dt; nwin += w; ++nbat;
				}
				{ std::lock_guard<std::mutex> lk(mu); freeb.push_back(B); }
				cv.notify_all();
			}
		};
		std::vector<std::thread> cons; for ( int g = 0; g < nwork; ++g ) cons.emplace_back(consumer);
		loader.join(); for ( size_t g = 0; g < cons.size(); ++g ) cons[g].join();
		double const tot = std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();
		if ( !fatal.empty() ) die(fatal);
		std::fprintf(stderr,"[L] host side only: %llu piles (%llu selected overlaps, %llu A-read bases, %llu windows) in %llu batches of %llu A reads, %.3f s wall\n"
			"[L] loader thread: %.3f s reading + selecting = %.0f piles/s = %.1f Mbase/s of A reads; %.3f s waiting for a free batch slot\n"
			"[L] %d planner thread(s) standing in for the device workers: %.3f s in dacc_plan_only in total = %.0f piles/s per worker\n",
			static_cast<unsigned long long>(npl),static_cast<unsigned long long>(nov),static_cast<unsigned long long>(abases),static_cast<unsigned long long>(nwin),
			static_cast<unsigned long long>(nbat),static_cast<unsigned long long>(o.batch),tot,
			load_s,load_s > 0 ? npl/load_s : 0.0,load_s > 0 ? abases/load_s/1e6 : 0.0,wait_s,
			nwork,plan_s,plan_s > 0 ? npl/plan_s : 0.0);
		return EXIT_SUCCESS;
	}

	// One context per device worker.  --gpus N uses the devices device, device+1, ...; when there are fewer devices the workers
	// wrap around (a legal, oversubscribed configuration: two contexts on one GPU overlap one batch's host plan with the other's
	// kernels; it is also how the mode is tested on a box with one GPU).  Piles are independent, so the batches are simply dealt to whichever worker is free; the writer
	// restores the batch order, and with it the reference's ascending A-read order and sequential well numbers.
	int const nwork = o.gpus;
	std::vector<dacc_ctx *> ctxs(nwork,static_cast<dacc_ctx *>(0));
	{
		// workers beyond the last device wrap around onto the devices from --device on; any OTHER failure of a worker's context
		// (out of memory, bad parameters) ends the run with its return code instead of silently oversubscribing a device
		int const ndev = dacc_device_count();
		if ( ndev <= 0 || o.device >= ndev ) die("no usable HIP device " + std::to_string(o.device) + " (" + std::to_string(ndev) + " visible)");
		int const span = ndev - o.device;
		for ( int g = 0; g < nwork; ++g )
		{
			dacc_params pg = p; pg.device = o.device + g % span;
			int const rc = dacc_create(&ctxs[g],&pg);
			if ( rc ) die("dacc_create failed (" + std::to_string(rc) + ") for worker " + std::to_string(g) + " on device " + std::to_string(pg.device) + " of " + std::to_string(ndev));
			if ( dacc_load_db(ctxs[g],pbps,nbps,pboff,prlen,nreads) ) die(std::string("load db: ") + dacc_last_error(ctxs[g]));
			if ( dacc_set_error_profile(ctxs[g],prof[0],prof[1],prof[2]) ) die(std::string("error profile: ") + dacc_last_error(ctxs[g]));
			if ( o.V && nwork > 1 ) std::fprintf(stderr,"[V] device worker %d on HIP device %d\n",g,pg.device);
		}
	}

	// Three stages, overlapped like the reference overlaps input, handlers and output with its threads (daccord.cpp:2107-2112):
	// a loader thread reads the byte range of the next batch of A reads from the .las and selects its piles, the device workers
	// plan their batches and run them on their GPUs, a writer thread turns the fragments into FASTA text in batch order.
	int64_t const batch = static_cast<int64_t>(o.batch);
	struct Out { std::vector<dacc_fragment> fr; std::string bases; uint64_t seq = 0; bool end = false; };
	std::mutex mu; std::condition_variable cv;
	std::deque<Batch *> loaded; std::deque<Batch *> freeb; std::deque<Out *> outs; std::deque<Out *> freeo;
	std::vector<Batch> bslots(2*nwork); std::vector<Out> oslots(2*nwork+1);
	for ( size_t i = 0; i < bslots.size(); ++i ) freeb.push_back(&bslots[i]);
	for ( size_t i = 0; i < oslots.size(); ++i ) freeo.push_back(&oslots[i]);
	auto const t0 = std::chrono::steady_clock::now();
	std::thread loader([&]() {
		uint64_t seq = 0;
		for ( int64_t b0 = minaread; ; b0 += batch )
		{
			Batch * B;
			{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk,[&]{ return !freeb.empty(); }); B = freeb.front(); freeb.pop_front(); }
			B->end = !(b0 < toparead); B->seq = seq++;
			if ( !B->end ) loadBatch(b0,std::min(toparead,b0+batch),false,*B);
			bool const stop = B->end || !B->err.empty();
			{ std::lock_guard<std::mutex> lk(mu); loaded.push_back(B); }
			cv.notify_all();
			if ( stop ) break;
		}
	});
	uint64_t well = 0, totalbases = 0;
	std::thread writer([&]() {
		std::string rec; uint64_t nextseq = 0;
		while ( true )
		{
			Out * O = 0;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk,[&]{ for ( size_t i = 0; i < outs.size(); ++i ) if ( outs[i]->seq == nextseq ) return true; return false; });
				for ( size_t i = 0; i < outs.size(); ++i ) if ( outs[i]->seq == nextseq ) { O = outs[i]; outs.erase(outs.begin()+i); break; }
			}
			++nextseq;
			if ( O->end ) break;
			rec.clear();
			for ( size_t i = 0; i < O->fr.size(); ++i )
			{
				dacc_fragment const & f = O->fr[i];
				char hdr[160];
				std::snprintf(hdr,sizeof(hdr),">%d/%llu/%u_%u A=[%u,%u]\n",f.aread+1,static_cast<unsigned long long>(well++),f.first,f.first+f.len,f.first,f.last);
				rec += hdr;
				for ( uint32_t q = 0; q < f.len; q += 80 ) { rec.append(O->bases.data()+f.seq_off+q,std::min<uint32_t>(80,f.len-q)); rec += '\n'; }
			}
			std::fwrite(rec.data(),1,rec.size(),stdout);
			{ std::lock_guard<std::mutex> lk(mu); freeo.push_back(O); }
			cv.notify_all();
		}
	});
	std::string fatal; bool stopall = false;           // both under mu
	double gpu_s = 0; uint64_t nbatches = 0;            // under mu
	// every batch (also an empty one, the end marker and a failed one) hands the writer an Out with its sequence number: the
	// writer needs an unbroken sequence.  An output slot is always available within bounded time: at most one Out per worker is
	// in flight besides those queued for the writer, and there are 2*nwork+1 slots.
	auto const deviceWorker = [&](int const g)
	{
		dacc_ctx * const ctx = ctxs[g];
		while ( true )
		{
			Batch * B = 0;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk,[&]{ return stopall || !loaded.empty(); });
				if ( stopall ) return;
				B = loaded.front();
				if ( B->end || !B->err.empty() ) stopall = true;      // the last batch: the other workers stop, this one reports it
				loaded.pop_front();
			}
			cv.notify_all();
			std::string err = B->err;
			bool const end = B->end; uint64_t const seq = B->seq;
			Out * O;
			{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk,[&]{ return !freeo.empty(); }); O = freeo.front(); freeo.pop_front(); }
			O->fr.clear(); O->bases.clear(); O->seq = seq; O->end = end || !err.empty();
			if ( err.empty() && !end && !B->spiles.empty() )
			{
				auto const tb0 = std::chrono::steady_clock::now();
				if ( dacc_submit_piles(ctx,B->spiles.data(),B->spiles.size(),B->sel.data(),B->sel.size(),B->trace.data(),B->ntrace,tbytes) )
					err = std::string("batch failed: ") + dacc_last_error(ctx);
				else
				{
					double const dt = std::chrono::duration<double>(std::chrono::steady_clock::now()-tb0).count();
					{ char const * pe = dacc_pile_errors(ctx); if ( pe && *pe ) std::fprintf(stderr,"[E] skipped reads:\n%s",pe); }
					dacc_fragment const * fr = 0; uint64_t nf = 0; char const * bases = 0; uint64_t nb = 0;
					if ( dacc_collect(ctx,&fr,&nf,&bases,&nb) ) err = std::string("collect: ") + dacc_last_error(ctx);
					else
					{
						O->fr.assign(fr,fr+nf); O->bases.assign(bases,nb);
						dacc_release(ctx);
						std::lock_guard<std::mutex> lk(mu); gpu_s += dt; ++nbatches; totalbases += nb;
					}
				}
			}
			bool last;
			{
				std::lock_guard<std::mutex> lk(mu);
				if ( !err.empty() ) { if ( fatal.empty() ) fatal = err; stopall = true; O->end = true; O->fr.clear(); O->bases.clear(); }
				last = O->end;      // (O belongs to the writer from here on)
				outs.push_back(O); freeb.push_back(B);
			}
			cv.notify_all();
			if ( last ) return;
		}
	};
	std::vector<std::thread> workers;
	for ( int g = 1; g < nwork; ++g ) workers.emplace_back(deviceWorker,g);
	deviceWorker(0);
	for ( size_t i = 0; i < workers.size(); ++i ) workers[i].join();
	// every batch taken from the loader is delivered (also a failed one, as an end marker), in particular all those in front of
	// an end marker: the writer always reaches it
	writer.join();
	if ( !fatal.empty() ) { std::fflush(stdout); std::fprintf(stderr,"[E] %s\n",fatal.c_str()); std::_Exit(EXIT_FAILURE); }
	loader.join();
	std::fflush(stdout);
	if ( o.V )
	{
		double const el = std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();
		std::fprintf(stderr,"[V] %llu corrected bases in %.2f s end to end (load + select + plan + GPU + FASTA) = %.3f Mbase/s; %llu batches, %.2f s in dacc_submit_piles%s\n",
			static_cast<unsigned long long>(totalbases),el,el > 0 ? totalbases/el/1e6 : 0.0,static_cast<unsigned long long>(nbatches),gpu_s,
			nwork > 1 ? (" (summed over " + std::to_string(nwork) + " device workers)").c_str() : "");
	}
	for ( int g = 0; g < nwork; ++g ) dacc_destroy(ctxs[g]);
	dacc_las_close(las); dacc_db_close(A.h); if ( twodb ) dacc_db_close(B2.h);
	return EXIT_SUCCESS;
}
