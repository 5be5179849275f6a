"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (flexpart_amd/).

CPU restatement of the reference's particle redistribution between MPI processes, /root/reference/src/mpi_mod.f90:
`mpif_calculate_part_redist` (:566-658) and `mpif_redist_part` (:661-856), in plain numpy / Python loops.

PARITY UNPINNED: mpi_mod.f90 needs an MPI library to compile and this image has none, so the restatement is not checked
against a build of the reference; it follows the source line by line (70 lines of index arithmetic, no floating point
besides the default-real comparison of the counts)."""
import numpy as np

MP_REDIST_FRACT = np.float32(0.2)     # mpi_mod.f90:156
MP_MIN_REDIST = 100000                # mpi_mod.f90:157
ARRAYS = ("nclass", "npoint", "itra1", "idt", "itramem", "itrasplit", "xtra1", "ytra1", "ztra1", "xmass1")


def plan(npart_per_process, ipout=1):
    """-> list of (src_proc, dest_proc, num_trans): the calls of mpif_redist_part, :633-655."""
    n = len(npart_per_process)
    if n == 1 or ipout == 3:                                         # :597, :613
        return []
    srt = [np.float32(c) for c in npart_per_process]                 # "sorted(:) = npart_per_process(:)": default reals
    idx = list(range(n))
    for i in range(0, n - 1):                                        # :616-631
        pmin, imin = srt[i], idx[i]
        for jj in range(i + 1, n):
            if pmin <= srt[jj]:
                continue
            pmin, srt[jj] = srt[jj], pmin
            imin, idx[jj] = idx[jj], imin
        srt[i], idx[i] = pmin, imin
    out = []
    m = n - 1
    for i in range(0, n // 2):                                       # :639-655
        hi, lo = int(npart_per_process[idx[m]]), int(npart_per_process[idx[i]])
        num_trans = hi - lo
        if hi > MP_MIN_REDIST and np.float32(num_trans) / np.float32(hi) > MP_REDIST_FRACT:
            out.append((idx[m], idx[i], num_trans // 2))
        m -= 1
    return out


def redist_part(src, dst, numpart_src, numpart_dst, num_trans, itime):
    """src, dst: dicts of the reference's particle arrays (ARRAYS; xmass1 [nspec][maxpart]) -- modified in place as the two
    processes of mpif_redist_part do.  -> (numpart_src, numpart_dst) afterwards."""
    ll, ul = numpart_src - num_trans, numpart_src                    # 0-based [ll, ul), :701-702
    tmp = {k: (np.array(src[k][:, ll:ul]) if k == "xmass1" else np.array(src[k][ll:ul])) for k in ARRAYS}
    src["itra1"][ll:ul] = -999999999                                 # :744
    numpart_src -= num_trans                                         # :746
    maxnumpart = numpart_dst + num_trans                             # :811
    minpart = 0
    ipart = None
    for i in range(num_trans):                                       # :815-834
        if tmp["itra1"][i] != itime:
            continue
        ipart = minpart
        while ipart < maxnumpart:
            if dst["itra1"][ipart] != itime:
                for k in ARRAYS:
                    if k == "xmass1":
                        dst[k][:, ipart] = tmp[k][:, i]
                    else:
                        dst[k][ipart] = tmp[k][i]
                break
            ipart += 1
        minpart = ipart + 1
    if ipart is not None:
        numpart_dst = max(numpart_dst, ipart + 1)                    # numpart=max(numpart,ipart), :836 (1-based ipart)
    return numpart_src, numpart_dst
