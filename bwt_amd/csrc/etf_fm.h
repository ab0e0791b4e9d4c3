// Reader / writer of the on-disk index of the reference:
//   <fm_indices>/<Chrom>_p<J>.fm = term_to_binary({[{pointers,{Pc,Pg,Pt,Last}},{shift,Shift}], FM})
// written by fm_index:make_index/4 (/root/reference/src/fm_index.erl:127-141) and read back by
// fm_index:get_index/2 (:154-156).  FM is a tuple of {F,L,I,SA} rows (src/bwt.erl:84-120).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

struct FmHost {
    std::vector<uint8_t> L;     // row r (1-based) at [r-1]
    std::vector<uint32_t> SA;
    uint32_t Pc = 0, Pg = 0, Pt = 0, Last = 0;
    int32_t shift = 0;
};

// Parses the external term format produced by OTP (uncompressed, tags 131/104/105/108/106/100/97/98,
// also 110/115/118/119 as later OTP releases write them).  F and I are checked against the identities
// they obey (F sorted = implied by the pointers; I[r] = C0[L[r]] + occ(L[r],r) - 1) and dropped.
// returns 0 or a negative BWT_E* code with text in err.
int fm_read_etf(const char *path, FmHost &out, std::string &err);

// Writes byte-for-byte what term_to_binary/1 would (OTP 17 atoms as ATOM_EXT).
int fm_write_etf(const char *path, const FmHost &fm, std::string &err);

// Derives {Pc,Pg,Pt,Last} from L (counts of each base; fmi:get_index_pointers/1, src/fmi.erl:66-76).
void fm_pointers_from_L(FmHost &fm);
