{-# LANGUAGE EmptyDataDecls           #-}
{-# LANGUAGE ForeignFunctionInterface #-}

-- | Raw FFI bindings to @libalchemy_hip.so@ (MI355X backend), one @foreign import@ per entry point of
-- @include/alchemy_hip.h@.  UNCOMPILED SOURCE: no Haskell toolchain exists in the pipeline that produced
-- this file; @tests/test_haskell_shim.py@ checks every import below against the header mechanically
-- (C name, arity, argument and result types).
--
-- Calling discipline (the same as lol-cpp's: caller-owned buffers, callee mutates in place, nothing retained):
--   * per-element Tensor methods take Lol's tuple-interleaved @Int64@ vectors (@Ptr Int64@);
--   * every function returns a status (@0@ = OK, @< 0@ = @ALCH_E_*@, message from 'c_lastError');
--     the @divG@ family returns @1@ for Lol's @Nothing@;
--   * calls that block on the device for more than microseconds are imported @safe@.
module Crypto.Lol.Cyclotomic.Tensor.GT.Backend where

import Data.Int
import Data.Word
import Foreign.C.String
import Foreign.C.Types
import Foreign.Ptr

data AlchRing
data AlchBuf
data AlchHint
data AlchTunnel

foreign import ccall unsafe "alch_last_error"          c_lastError       :: IO CString
foreign import ccall unsafe "alch_version"             c_version         :: IO Word32

-- ring context: one per (index m, modulus list); q ≡ 1 (mod m) or ALCH_E_NO_CRT (Lol: crtFuncs = Nothing)
foreign import ccall safe   "alch_ring_create"         c_ringCreate      :: Word32 -> CInt -> Ptr Word64 -> Ptr (Ptr AlchRing) -> IO CInt
foreign import ccall safe   "alch_ring_create_nocrt"   c_ringCreateNoCRT :: Word32 -> CInt -> Ptr Word64 -> Ptr (Ptr AlchRing) -> IO CInt
foreign import ccall safe   "alch_ring_destroy"        c_ringDestroy     :: Ptr AlchRing -> IO CInt
foreign import ccall unsafe "alch_select_limbs"        c_selectLimbs     :: Ptr Word64 -> CInt -> CInt -> CInt -> CInt -> Ptr CInt -> Ptr CInt -> Ptr CInt -> Ptr CInt -> IO CInt
foreign import ccall unsafe "alch_modulus_units"       c_modulusUnits    :: Word64 -> IO CInt
foreign import ccall unsafe "alch_host_root"           c_hostRoot        :: Word32 -> Word64 -> Ptr Word64 -> Ptr Word64 -> IO CInt
foreign import ccall unsafe "alch_ring_n"              c_ringN           :: Ptr AlchRing -> Ptr Word32 -> Ptr CInt -> Ptr CInt -> IO CInt
foreign import ccall safe   "alch_ring_set_stream"     c_ringSetStream   :: Ptr AlchRing -> Ptr () -> IO CInt
foreign import ccall unsafe "alch_buf_ring"            c_bufRing         :: Ptr AlchBuf -> Ptr (Ptr AlchRing) -> IO CInt
foreign import ccall unsafe "alch_ring_device"         c_ringDevice      :: Ptr AlchRing -> Ptr CInt -> Ptr (Ptr ()) -> IO CInt
foreign import ccall safe   "alch_ring_share_stream"   c_ringShareStream :: Ptr AlchRing -> Ptr AlchRing -> IO CInt
foreign import ccall safe   "alch_ring_set_option"     c_ringSetOption   :: Ptr AlchRing -> CString -> CLong -> IO CInt
foreign import ccall safe   "alch_sync"                c_sync            :: Ptr AlchRing -> IO CInt
foreign import ccall safe   "alch_timer_start"         c_timerStart      :: Ptr AlchRing -> IO CInt
foreign import ccall safe   "alch_timer_stop"          c_timerStop       :: Ptr AlchRing -> Ptr CFloat -> IO CInt

-- Tensor crt / crtInv (Lol: crtFuncs), zipWithT (*) (+) (-), scalar multiply: one ring element, in place
foreign import ccall safe   "alch_crt"                 c_crt             :: Ptr AlchRing -> Ptr Int64 -> IO CInt
foreign import ccall safe   "alch_crtinv"              c_crtInv          :: Ptr AlchRing -> Ptr Int64 -> IO CInt
foreign import ccall safe   "alch_mul"                 c_mul             :: Ptr AlchRing -> Ptr Int64 -> Ptr Int64 -> IO CInt
foreign import ccall safe   "alch_add"                 c_add             :: Ptr AlchRing -> Ptr Int64 -> Ptr Int64 -> IO CInt
foreign import ccall safe   "alch_sub"                 c_sub             :: Ptr AlchRing -> Ptr Int64 -> Ptr Int64 -> IO CInt
foreign import ccall safe   "alch_scale"               c_scale           :: Ptr AlchRing -> Ptr Int64 -> Ptr Word64 -> IO CInt

-- mulGPow/Dec/CRT, divGPow/Dec/CRT (divG: 1 = Nothing), l / lInv
foreign import ccall safe   "alch_mulg_pow"            c_mulGPow         :: Ptr AlchRing -> Ptr Int64 -> IO CInt
foreign import ccall safe   "alch_mulg_dec"            c_mulGDec         :: Ptr AlchRing -> Ptr Int64 -> IO CInt
foreign import ccall safe   "alch_mulg_crt"            c_mulGCRT         :: Ptr AlchRing -> Ptr Int64 -> IO CInt
foreign import ccall safe   "alch_divg_pow"            c_divGPow         :: Ptr AlchRing -> Ptr Int64 -> IO CInt
foreign import ccall safe   "alch_divg_dec"            c_divGDec         :: Ptr AlchRing -> Ptr Int64 -> IO CInt
foreign import ccall safe   "alch_divg_crt"            c_divGCRT         :: Ptr AlchRing -> Ptr Int64 -> IO CInt
foreign import ccall safe   "alch_l"                   c_l               :: Ptr AlchRing -> Ptr Int64 -> IO CInt
foreign import ccall safe   "alch_linv"                c_lInv            :: Ptr AlchRing -> Ptr Int64 -> IO CInt

-- Tensor methods between two indices m | m': embedPow / embedDec / twacePowDec / coeffs and crtExtFuncs = (twaceCRT, embedCRT);
-- host-only index tables (powBasisPow) and crtSetDec
foreign import ccall safe   "alch_embed_pow"           c_embedPow        :: Ptr AlchRing -> Ptr AlchRing -> Ptr Int64 -> Ptr Int64 -> IO CInt
foreign import ccall safe   "alch_embed_dec"           c_embedDec        :: Ptr AlchRing -> Ptr AlchRing -> Ptr Int64 -> Ptr Int64 -> IO CInt
foreign import ccall safe   "alch_embed_crt"           c_embedCRT        :: Ptr AlchRing -> Ptr AlchRing -> Ptr Int64 -> Ptr Int64 -> IO CInt
foreign import ccall safe   "alch_twace_pow_dec"       c_twacePowDec     :: Ptr AlchRing -> Ptr AlchRing -> Ptr Int64 -> Ptr Int64 -> IO CInt
foreign import ccall safe   "alch_twace_crt"           c_twaceCRT        :: Ptr AlchRing -> Ptr AlchRing -> Ptr Int64 -> Ptr Int64 -> IO CInt
foreign import ccall safe   "alch_coeffs"              c_coeffs          :: Ptr AlchRing -> Ptr AlchRing -> Ptr Int64 -> Ptr Int64 -> IO CInt
foreign import ccall unsafe "alch_ext_table"           c_extTable        :: Word32 -> Word32 -> CInt -> Ptr Int32 -> Ptr CSize -> IO CInt
foreign import ccall safe   "alch_crt_set_dec"         c_crtSetDec       :: Word32 -> Word32 -> Word32 -> Ptr Int64 -> Ptr CSize -> IO CInt

-- decompose + reduce (Gadget / Decompose instances of TrivGad and BaseBGad 2)
foreign import ccall safe   "alch_decompose_triv"      c_decomposeTriv   :: Ptr AlchRing -> Ptr Int64 -> Ptr Int64 -> IO CInt
foreign import ccall safe   "alch_decompose_base2"     c_decomposeBase2  :: Ptr AlchRing -> Ptr Int64 -> Ptr Int64 -> Ptr CInt -> IO CInt

-- device-resident batches
foreign import ccall safe   "alch_buf_alloc"           c_bufAlloc        :: Ptr AlchRing -> CSize -> Ptr (Ptr AlchBuf) -> IO CInt
foreign import ccall safe   "alch_buf_free"            c_bufFree         :: Ptr AlchBuf -> IO CInt
-- device-resident Tensor values (GT's constructor GTDev): pooled single-element buffers, aliases, copies, the unary Tensor methods
-- out of place.  Everything that touches a ring takes the device lock (include/alchemy_hip.h, threading) and may wait behind another
-- thread's call: `safe` imports; only the pure accessors below stay `unsafe`.
foreign import ccall unsafe "alch_buf_view"            c_bufView         :: Ptr AlchBuf -> CSize -> CSize -> Ptr (Ptr AlchBuf) -> IO CInt
foreign import ccall safe   "alch_buf_copy"            c_bufCopy         :: Ptr AlchBuf -> CSize -> Ptr AlchBuf -> CSize -> CSize -> IO CInt
foreign import ccall safe   "alch_buf_tensor_op"       c_bufTensorOp     :: Ptr AlchBuf -> CSize -> Ptr AlchBuf -> CSize -> CSize -> CInt -> IO CInt
foreign import ccall unsafe "alch_buf_elems"           c_bufElems        :: Ptr AlchBuf -> Ptr CSize -> IO CInt
foreign import ccall unsafe "alch_buf_device_ptr"      c_bufDevicePtr    :: Ptr AlchBuf -> Ptr (Ptr ()) -> Ptr CSize -> IO CInt
foreign import ccall safe   "alch_buf_upload"          c_bufUpload       :: Ptr AlchBuf -> CSize -> CSize -> Ptr Int64 -> IO CInt
foreign import ccall safe   "alch_buf_download"        c_bufDownload     :: Ptr AlchBuf -> CSize -> CSize -> Ptr Int64 -> IO CInt
foreign import ccall safe   "alch_buf_fill_uniform"    c_bufFillUniform  :: Ptr AlchBuf -> Word64 -> IO CInt
foreign import ccall safe   "alch_buf_crt"             c_bufCrt          :: Ptr AlchBuf -> CSize -> CSize -> IO CInt
foreign import ccall safe   "alch_buf_crtinv"          c_bufCrtInv       :: Ptr AlchBuf -> CSize -> CSize -> IO CInt
foreign import ccall safe   "alch_buf_mul"             c_bufMul          :: Ptr AlchBuf -> Ptr AlchBuf -> Ptr AlchBuf -> CSize -> IO CInt
foreign import ccall safe   "alch_buf_add"             c_bufAdd          :: Ptr AlchBuf -> Ptr AlchBuf -> Ptr AlchBuf -> CSize -> IO CInt
foreign import ccall safe   "alch_buf_sub"             c_bufSub          :: Ptr AlchBuf -> Ptr AlchBuf -> Ptr AlchBuf -> CSize -> IO CInt
foreign import ccall safe   "alch_buf_scale"           c_bufScale        :: Ptr AlchBuf -> Ptr AlchBuf -> CSize -> Ptr Word64 -> IO CInt
foreign import ccall safe   "alch_buf_decompose_triv"  c_bufDecomposeTriv :: Ptr AlchBuf -> CSize -> Ptr AlchBuf -> CSize -> IO CInt
foreign import ccall safe   "alch_buf_l"               c_bufL            :: Ptr AlchBuf -> CSize -> CSize -> IO CInt
foreign import ccall safe   "alch_buf_linv"            c_bufLInv         :: Ptr AlchBuf -> CSize -> CSize -> IO CInt
foreign import ccall safe   "alch_buf_mulg"            c_bufMulG         :: Ptr AlchBuf -> CSize -> CSize -> CInt -> IO CInt
foreign import ccall safe   "alch_buf_divg"            c_bufDivG         :: Ptr AlchBuf -> CSize -> CSize -> CInt -> IO CInt
foreign import ccall safe   "alch_buf_mul_public"      c_bufMulPublic    :: Ptr AlchBuf -> Ptr AlchBuf -> Ptr AlchBuf -> CSize -> CSize -> IO CInt
foreign import ccall safe   "alch_buf_add_public"      c_bufAddPublic    :: Ptr AlchBuf -> Ptr AlchBuf -> CSize -> CSize -> IO CInt
foreign import ccall safe   "alch_buf_embed"           c_bufEmbed        :: Ptr AlchBuf -> Ptr AlchBuf -> CSize -> CInt -> IO CInt
foreign import ccall safe   "alch_buf_twace"           c_bufTwace        :: Ptr AlchBuf -> Ptr AlchBuf -> CSize -> CInt -> IO CInt
foreign import ccall safe   "alch_buf_coeffs"          c_bufCoeffs       :: Ptr AlchBuf -> Ptr AlchBuf -> CSize -> IO CInt
foreign import ccall safe   "alch_ct_add_public"       c_ctAddPublic     :: Ptr AlchBuf -> Ptr AlchBuf -> CSize -> Ptr Word64 -> Ptr AlchBuf -> CSize -> IO CInt
foreign import ccall safe   "alch_buf_checksum"        c_bufChecksum     :: Ptr AlchBuf -> CSize -> CSize -> Ptr Word64 -> IO CInt
foreign import ccall safe   "alch_buf_checksum_at"     c_bufChecksumAt   :: Ptr AlchBuf -> CSize -> CSize -> Word64 -> Ptr Word64 -> IO CInt
foreign import ccall safe   "alch_buf_rescale_drop0"   c_bufRescaleDrop0 :: Ptr AlchBuf -> Ptr AlchBuf -> CSize -> IO CInt
foreign import ccall safe   "alch_buf_rescale_add0"    c_bufRescaleAdd0  :: Ptr AlchBuf -> Ptr AlchBuf -> CSize -> IO CInt

-- KSQuadCircHint (gadget: 0 = TrivGad, 1 = BaseBGad 2)
foreign import ccall safe   "alch_hint_load"           c_hintLoad        :: Ptr AlchRing -> CInt -> Ptr Int64 -> Ptr (Ptr AlchHint) -> IO CInt
foreign import ccall safe   "alch_hint_from_buf"       c_hintFromBuf     :: Ptr AlchRing -> CInt -> Ptr AlchBuf -> Ptr (Ptr AlchHint) -> IO CInt
foreign import ccall safe   "alch_hint_free"           c_hintFree        :: Ptr AlchHint -> IO CInt

-- ring tunnelling (SymmSHE tunnel between PT2CT's two modSwitch_)
foreign import ccall unsafe "alch_tunnel_info"         c_tunnelInfo      :: Ptr AlchRing -> Ptr AlchRing -> Ptr Word32 -> Ptr Word32 -> IO CInt
foreign import ccall safe   "alch_tunnel_create"       c_tunnelCreate    :: Ptr AlchRing -> Ptr AlchRing -> CInt -> Ptr AlchBuf -> Ptr AlchBuf -> Ptr (Ptr AlchTunnel) -> IO CInt
foreign import ccall safe   "alch_tunnel_free"         c_tunnelFree      :: Ptr AlchTunnel -> IO CInt
foreign import ccall safe   "alch_ct_tunnel"           c_ctTunnel        :: Ptr AlchTunnel -> Ptr AlchBuf -> Ptr AlchBuf -> CSize -> Ptr Word64 -> CUInt -> IO CInt

foreign import ccall safe   "alch_ct_mod_switch"       c_ctModSwitch     :: Ptr AlchBuf -> Ptr AlchBuf -> CSize -> CUInt -> IO CInt

-- the hot path: keySwitchQuadCirc hint (a * b), and PT2CT's whole mul_
foreign import ccall safe   "alch_ct_mul_relin"        c_ctMulRelin      :: Ptr AlchRing -> Ptr AlchHint -> Ptr AlchBuf -> Ptr AlchBuf -> Ptr AlchBuf -> CSize -> Ptr Word64 -> CUInt -> IO CInt
foreign import ccall safe   "alch_ct_mul_full"         c_ctMulFull       :: Ptr AlchHint -> Ptr AlchBuf -> Ptr AlchBuf -> Ptr AlchBuf -> CSize -> Ptr Word64 -> CUInt -> IO CInt
