#!/usr/bin/env python3
"""Build sbdart_amd/data/sbdart_tables.bin from the COMPILED reference (build-container tool).

The band model needs the reference's physical DATA tables: LOWTRAN7 band-model
coefficients, continuum cross-sections, trace-gas mixing ratios, the six model atmospheres, the solar
spectra.  They are data, not design; none of the reference's source text is read or kept.  Two
mechanisms, both against oracle/_ref/libsbdart_ref.so (oracle/build_ref.sh compiles it from the
sources where they lie):

  * static tables: the bytes of the array's symbol in the library image (ELF .symtab -> section ->
    file offset).  Values are exactly what the compiler stored, i.e. fp32 literals widened to fp64
    where the reference wrote them that way;
  * tables the reference builds at run time (model atmospheres, solar spectra on their wavelength
    grids): the reference routine is called through ctypes and its output arrays are stored.

File format (little-endian): "SBDTBL1\\0", int32 ntab, then per table: char[24] name, int32 kind
(1 = float64, 2 = int32), int32 n, n values, padded to a multiple of 8 bytes.  A manifest with the
provenance of every table goes beside it (TABLES.md).  Single reader: sbd_tables_mod.f90.
"""
import ctypes
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(ROOT, "oracle", "_ref", "libsbdart_ref.so")
OUT = os.path.join(ROOT, "sbdart_amd", "data", "sbdart_tables.bin")
MXLY = 65


class Elf:
    """The few ELF64 facts needed: symbol name -> (address, size), address -> file bytes."""

    def __init__(self, path):
        self.b = open(path, "rb").read()
        b = self.b
        assert b[:4] == b"\x7fELF" and b[4] == 2 and b[5] == 1, "ELF64 little-endian expected"
        shoff, = struct.unpack_from("<Q", b, 0x28)
        shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, 0x3A)
        self.sections = []
        for i in range(shnum):
            (name, typ, flags, addr, off, size, link, info, align, entsize) = struct.unpack_from(
                "<IIQQQQIIQQ", b, shoff + i * shentsize)
            self.sections.append(dict(name=name, type=typ, addr=addr, off=off, size=size, link=link, entsize=entsize))
        self.syms = {}
        for s in self.sections:
            if s["type"] != 2:          # SHT_SYMTAB
                continue
            strtab = self.sections[s["link"]]
            for k in range(s["size"] // s["entsize"]):
                st_name, st_info, st_other, st_shndx, st_value, st_size = struct.unpack_from(
                    "<IBBHQQ", b, s["off"] + k * s["entsize"])
                end = b.index(b"\0", strtab["off"] + st_name)
                nm = b[strtab["off"] + st_name:end].decode()
                if nm and st_size:
                    self.syms[nm] = (st_value, st_size)

    def data(self, symbol, dtype):
        addr, size = self.syms[symbol]
        for s in self.sections:
            if s["type"] != 8 and s["addr"] <= addr and addr + size <= s["addr"] + s["size"] and s["addr"]:
                off = addr - s["addr"] + s["off"]
                return np.frombuffer(self.b[off:off + size], dtype=dtype).copy()
        raise KeyError("%s is not in an initialised section" % symbol)


def main():
    if not os.path.exists(LIB):
        sys.exit("extract_tables: %s missing -- run oracle/build_ref.sh in the build container" % LIB)
    elf = Elf(LIB)
    lib = ctypes.CDLL(LIB)
    tables = []          # (name, array, provenance)

    def static(name, symbol, where, dtype="<f8"):
        tables.append((name, elf.data(symbol, dtype), "image of `%s` (%s)" % (symbol, where)))

    # ---- LOWTRAN7 band-model absorption coefficients, 5 cm-1 steps (module gasblk, taugas.f:13-1798) ----
    for mol in ("h2o", "o3", "co2", "co", "ch4", "n2o", "o2", "nh3", "no", "no2", "so2"):
        static("cp." + mol, "_QMgasblkEcp" + mol, "taugas.f:19-1798")
        static("iwl." + mol, "_QMgasblkEiwl" + mol, "taugas.f:23-28", "<i4")
        static("iwh." + mol, "_QMgasblkEiwh" + mol, "taugas.f:23-28", "<i4")
    # ---- band parameters per absorber band (abcdta, taugas.f:6458-6735) ----
    for mol in ("h2o", "o3", "co2", "co", "n2o", "o2", "nh3", "so2"):
        for pre, key in (("a", "bs"), ("aa", "ba"), ("bb", "bb"), ("cc", "bc")):
            static("%s.%s" % (key, mol), "_QFabcdtaE%s%s" % (pre, mol), "taugas.f:6475-6540")
    # NO has one band, CH4 and NO2 have the same parameters in each of their bands: the compiler folded
    # these DATA values into the code, so there is no array image.  They are the fp32 literals of
    # taugas.f:6478-6521 widened to fp64, exactly as the reference's DATA statements produce them.
    for mol, nband, lits in (("no", 1, (".6613", ".083336", ".319585", "34.6834")),
                             ("ch4", 4, (".5844", ".154447", ".357657", "25.8920")),
                             ("no2", 3, (".7249", ".045281", ".264248", "42.2784"))):
        for key, lit in zip(("bs", "ba", "bb", "bc"), lits):
            tables.append(("%s.%s" % (key, mol), np.full(nband, np.float64(np.float32(lit))),
                           "fp32 literal %s widened, %d band(s) (DATA in abcdta, taugas.f:6478-6521)" % (lit, nband)))
    # ---- continua ----
    static("h2o.self296", "_QFslf296Es", "taugas.f:2538-2975")
    static("h2o.self260", "_QFslf260Es", "taugas.f:2977-3414")
    static("h2o.foreign", "_QFfrn296Ef", "taugas.f:3416-3852")
    static("n2.cont", "_QFc4dtaEc4", "taugas.f:3873-3916")
    for k in (1, 2, 3):
        static("hno3.h%d" % k, "_QFhno3Eh%d" % k, "taugas.f:3918-3951")
    static("o2.s0", "_QFo2contEo2s0", "taugas.f:3995-4150")
    static("o2.a", "_QFo2contEo2a", "taugas.f:3995-4150")
    static("o2.b", "_QFo2contEo2b", "taugas.f:3995-4150")
    for k in (0, 1, 2):
        static("o3.hh%d" % k, "_QFo3hhtEs%d" % k, "taugas.f:4152-6302")
    static("o3.uv", "_QFo3uvEs", "taugas.f:6304-6369")
    static("o3.chappuis", "_QFc8dtaEc8", "taugas.f:6371-6414")
    static("o2.schrun", "_QFschrunEshn", "taugas.f:6736-6821")
    static("o4.sig", "_QFo4contEsig", "taugas.f:6940-7176")
    # ---- trace-gas mixing-ratio profiles (module trcblk, taugas.f:6823-6938) ----
    for g in ("alt", "n2", "o2", "co2", "ch4", "n2o", "co", "no2", "so2", "nh3", "no", "hno3"):
        static("mix." + g, "_QMtrcblkE" + g, "taugas.f:6834-6937")

    # ---- clouds: Mie efficiency / single-scattering albedo / asymmetry on 400 log-spaced wavelengths x
    #      13 effective radii, water and ice (cloudpar, taucloud.f:344-6768); tabulated phase-function
    #      moments of GETMOM (disutil.f:2104-2209) ----
    for key, sym in (("q", "qq"), ("w", "ww"), ("g", "gg"), ("qi", "qqi"), ("wi", "wwi"), ("gi", "ggi")):
        static("cloud." + key, "_QFcloudparE" + sym, "taucloud.f:398-6725, [400 wavelengths][13 radii] column-major")
    static("pmom.haze_l", "_QFgetmomEhazelm", "disutil.f:2109-2124")
    static("pmom.cloud_c1", "_QFgetmomEcldmom", "disutil.f:2125-2162")

    # ---- model atmospheres: the reference's profile routines, called (atms.f:660-1068) ----
    dp = ctypes.POINTER(ctypes.c_double)
    for idatm, fn in enumerate(("tropic_", "midsum_", "midwin_", "subsum_", "subwin_", "us62_"), start=1):
        nz = ctypes.c_int(0)
        arr = [np.zeros(MXLY) for _ in range(5)]
        getattr(lib, fn)(ctypes.byref(nz), *[a.ctypes.data_as(dp) for a in arr])
        n = nz.value
        assert 2 <= n <= MXLY
        tables.append(("atm%d" % idatm, np.concatenate([a[:n] for a in arr]),
                       "output of `%s` (z, p, t, wh, wo; %d levels bottom-up; atms.f:660-1068)" % (fn, n)))
    # ---- solar spectra on their own wavelength grids (spectra.f:1417-3238) ----
    for nf, fn in ((1, "sun1s_"), (2, "sunlow_"), (3, "sunmod_")):
        nns = ctypes.c_int(5000)
        wl, irr = np.zeros(8192), np.zeros(8192)
        getattr(lib, fn)(wl.ctypes.data_as(dp), irr.ctypes.data_as(dp), ctypes.byref(nns))
        n = nns.value
        tables.append(("sun%d.wl" % nf, wl[:n].copy(), "output of `%s` (spectra.f:1417-3238)" % fn))
        tables.append(("sun%d.irr" % nf, irr[:n].copy(), "output of `%s`" % fn))

    # ---- aerosols (tauaero.f): wavelengths of the standard models, extinction / absorption / asymmetry of
    #      the rural, urban, oceanic and tropospheric boundary-layer models at four relative humidities,
    #      the four stratospheric models, and the standard vertical profile (the profile aerzstd leaves
    #      in the module: the reference's visibility weighting always selects the 23 km profile) ----
    static("aer.wl", "_QMaeroblkEawl", "tauaero.f:47-58")
    static("aer.rhzone", "_QMaeroblkFstdaerErhzone", "tauaero.f:600")
    for model, pre in (("rural", "rur"), ("urban", "urb"), ("ocean", "ocn"), ("tropo", "tro")):
        for q in "eag":
            static("aer.%s.%s" % (model, q), "_QMaeroblkFstdaerE%s%s" % (pre, q),
                   "tauaero.f:610-1010, [47 wavelengths][4 humidities] column-major")
    static("aer.strat", "_QMaeroblkFaestratEaerstr",
           "tauaero.f:262-395, [47 wavelengths][ext, abs, asym][4 models] column-major")
    static("aer.z", "_QMaeroblkFaerzstdEalt", "tauaero.f:93-101")
    ctypes.c_double.in_dll(lib, "_QMaeroblkEvis").value = 23.0
    getattr(lib, "_QMaeroblkPaerzstd")()
    dens = np.array((ctypes.c_double * 33).in_dll(lib, "_QMaeroblkEdbaer"))
    zchk = np.array((ctypes.c_double * 33).in_dll(lib, "_QMaeroblkEzbaer"))
    assert np.array_equal(zchk, tables[-1][1])
    tables.append(("aer.density", dens, "module array `dbaer` after `aerzstd` (tauaero.f:88-140)"))

    # ---- spectral albedo of the six standard surfaces on their own grids (spectra.f:2899-3238) ----
    for isalb, fn in enumerate(("snow_", "clearw_", "lakew_", "seaw_", "sand_", "vegeta_"), start=1):
        nna = ctypes.c_int(5000)
        wl, alb = np.zeros(8192), np.zeros(8192)
        getattr(lib, fn)(wl.ctypes.data_as(dp), alb.ctypes.data_as(dp), ctypes.byref(nna))
        n = nna.value
        tables.append(("alb%d.wl" % isalb, wl[:n].copy(), "output of `%s` (spectra.f:2899-3238)" % fn))
        tables.append(("alb%d.r" % isalb, alb[:n].copy(), "output of `%s`" % fn))

    # ---- equation of time (minutes) and solar declination (degrees) every five days (zensun, spectra.f:4440-4556)
    static("sun.eqt", "_QFzensunEeqt", "spectra.f:4467-4477")
    static("sun.dec", "_QFzensunEdec", "spectra.f:4478-4488")

    # ---- ocean surface (ISALB 7, seabdrf): refractive index of water (indwat, spectra.f:594-1222) and Morel's case-I
    #      water coefficients on 400..700 nm by 5 nm (morcasiwat, spectra.f:467-592) ----
    static("ocean.wl", "_QFindwatEwltab", "spectra.f:623-800")
    static("ocean.mr", "_QFindwatEmrtab", "spectra.f:801-985")
    static("ocean.mi", "_QFindwatEmitab", "spectra.f:986-1190")
    static("ocean.kw", "_QFmorcasiwatEtkw", "spectra.f:499-513")
    static("ocean.xc", "_QFmorcasiwatEtxc", "spectra.f:514-528")
    static("ocean.e", "_QFmorcasiwatEte", "spectra.f:529-543")
    static("ocean.bw", "_QFmorcasiwatEtbw", "spectra.f:544-558")

    # ---- sensor response functions ISAT 1..29 on their even wavelength grids (spectra.f:3414-4380):
    #      [wlmin, wlmax, response(1:n)] ----
    sensors = ("meteo", "goese", "goesw", "avhr81", "avhr82", "avhr91", "avhr92", "avhr101", "avhr102", "avhr111",
               "avhr112", "gtr1", "gtr2", "nm410", "nm936", "mfrsr1", "mfrsr2", "mfrsr3", "mfrsr4", "mfrsr5",
               "mfrsr6", "avhr83", "avhr84", "avhr85", "setlow", "airs1", "airs2", "airs3", "airs4")
    for isat, fn in enumerate(sensors, start=1):
        nnf = ctypes.c_int(0)
        wmin, wmax = ctypes.c_double(0), ctypes.c_double(0)
        resp = np.zeros(8192)
        getattr(lib, fn + "_")(resp.ctypes.data_as(dp), ctypes.byref(wmin), ctypes.byref(wmax), ctypes.byref(nnf))
        n = nnf.value
        assert 2 <= n <= 5000, (fn, n)
        tables.append(("filter%d" % isat, np.concatenate([[wmin.value, wmax.value], resp[:n]]),
                       "output of `%s_`: wlmin, wlmax, %d response values (spectra.f:3414-4380)" % (fn, n)))

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "wb") as f:
        f.write(b"SBDTBL1\0")
        f.write(struct.pack("<i", len(tables)))
        for name, a, _ in tables:
            assert len(name) <= 24
            kind = 2 if a.dtype.kind == "i" else 1
            a = a.astype("<i4" if kind == 2 else "<f8")
            f.write(name.encode().ljust(24, b" "))
            f.write(struct.pack("<ii", kind, a.size))
            raw = a.tobytes()
            f.write(raw + b"\0" * (-len(raw) % 8))
    with open(os.path.join(os.path.dirname(OUT), "TABLES.md"), "w") as f:
        f.write("# sbdart_tables.bin -- provenance\n\n"
                "Generated by `tools/extract_tables.py` from the reference compiled by `oracle/build_ref.sh`\n"
                "(`oracle/_ref/libsbdart_ref.so`); physical data of LOWTRAN7 / 5S / MODTRAN3 and the AFGL model\n"
                "atmospheres as the reference ships them.  `file:line` are into paulricchiazzi/SBDART.\n\n"
                "| table | type | n | source |\n|---|---|---|---|\n")
        for name, a, prov in tables:
            f.write("| `%s` | %s | %d | %s |\n" % (name, "i32" if a.dtype.kind == "i" else "f64", a.size, prov))
    print("wrote %s: %d tables, %d bytes" % (OUT, len(tables), os.path.getsize(OUT)))


if __name__ == "__main__":
    main()
