{-# LANGUAGE ConstraintKinds       #-}
{-# LANGUAGE DataKinds             #-}
{-# LANGUAGE FlexibleContexts      #-}
{-# LANGUAGE FlexibleInstances     #-}
{-# LANGUAGE GADTs                 #-}
{-# LANGUAGE KindSignatures        #-}
{-# LANGUAGE MultiParamTypeClasses #-}
{-# LANGUAGE PolyKinds             #-}
{-# LANGUAGE RankNTypes            #-}
{-# LANGUAGE ScopedTypeVariables   #-}
{-# LANGUAGE TypeFamilies          #-}
{-# LANGUAGE TypeOperators         #-}
{-# LANGUAGE UndecidableInstances  #-}

-- | @GT@: a Lol 'Tensor' whose values can live in the HBM of an MI355X and whose basis-order-dependent methods run in the
-- library declared by @include/alchemy_hip.h@.
--
-- UNCOMPILED SOURCE.  No Haskell toolchain (and no Lol) exists in the pipeline that produced this file: it is written against the
-- Lol 0.7 @Tensor@ class from its published interface and checked mechanically only (@tests/test_haskell_shim.py@: every class
-- method is defined, none is a stub, every foreign symbol exists in @GT/Backend.hs@ with the header's signature, every status of
-- @alch_ring_create@ that stands for a Lol @Nothing@ has a non-@error@ branch, the op codes equal the header's).  What IS compiled
-- and tested is the same design one level down: @alchemy_amd/host/cycgen.hpp@ (C++; modes @HostBuffers@ = 'GTHost', @Resident@ =
-- 'GTDev', @ResidentZipHost@ = 'GTDev' under an unchanged Lol), replaying the reference's examples to their @PASS@, and
-- @tests/test_gpu_example_rings.py@, which follows 'ringFor' / 'powRing' / 'crtFuncsGT' / 'crtExtFuncsGT' call for call through
-- the C ABI on every ring the three examples instantiate.  Places where a first compile is most likely to want an edit: the
-- exact superclass contexts of the @entail*@ instances at the end, and lol-cpp's internal module name for @CT@'s constructors.
--
-- How the method bodies are written.  Class-method type variables are not in scope in an instance body, and the exact contexts of
-- Lol's methods are not known here, so no method body names a type variable or adds a constraint: each one hands lol-cpp's
-- implementation of ITSELF (under the 'GTHost' constructor, typed by inference) to a top-level helper whose own signature binds
-- the indices -- @twacePowDec = twacePowDecGT (liftH twacePowDec)@ -- and the helper needs only @TElt GT r@ and the index
-- constraint, which every method of the class certainly provides.
--
-- Representation.  @GT m r@ is either a host vector (lol-cpp's @CT m r@) or ONE ring element resident in HBM
-- ('GTDev': a pooled @alch_buf@ of the ring that serves @(m, r)@).  @E@ issues one Lol call per op (reference
-- Crypto/Alchemy/Interpreter/Eval.hs:120-134) and Lol one Tensor call per basis change, so an operation arrives here as a chain
-- of single-element calls; a 'GTDev' result stays on the device for the next link (one asynchronous kernel launch per call, no
-- PCIe crossing, no synchronisation) and is downloaded only where Lol needs host data: 'fmapT', 'zipWithT', 'unzipT', @Show@,
-- @Eq@, @gSqNormDec@, the entailed instances.
--
-- Which (index, element type) pairs are served.  Element types with moduli -- (nested pairs of) @ZqBasic q Int64@, 'GTDispatch' --
-- on an index the library serves ('powRing': the ring with CRT basis when every q is a prime = 1 mod m, else a ring without CRT
-- basis: the plaintext rings @Z_{2^e}@ of reference examples/Common.hs:32 and @Zq 7@ over @F4@ of examples/Arithmetic.hs:23).
-- For a served pair EVERY method whose result depends on the order of a basis crosses into the library -- @l@, @lInv@,
-- @mulGPow/Dec@, @divGPow/Dec@, @crtFuncs@ (@crt@, @crtInv@, @mulGCRT@, @divGCRT@), @crtExtFuncs@ (@twaceCRT@, @embedCRT@),
-- @twacePowDec@, @embedPow@, @embedDec@, @coeffs@, @powBasisPow@, @crtSetDec@ -- because the CRT slot order of an instance is
-- internal to it but must be ONE order.  Every other pair (@Double@, @Complex Double@, @Int64@, an index with a prime factor above
-- 13 or a limb-polynomial beyond the LDS: @ALCH_E_UNSUPPORTED@) is lol-cpp's, also as a whole.  The one mixed case -- an extension
-- m | m' whose small index is served and whose big index is not -- composes through the powerful basis, which both sides index the
-- same way (ASSUMPTION, unpinned like the rest of the parity story: the Pow / Dec orders are Lol's documented mixed-radix orders,
-- which this library follows; they must agree anyway because @Double@ / @Int64@ tensors and @Zq@ tensors meet in @tGaussianDec@,
-- @lift@ and @reduce@): @embedCRT = crt' . embedPow . crtInv@, @twaceCRT = crt . twacePowDec . crtInv'@.
--
-- Pointwise ring operations.  Lol's @UCyc@ multiplies and adds with @zipWithT f@ for an opaque @f@, which cannot be shipped to the
-- GPU; GHC rewrite rules on class methods do not fire at Lol's polymorphic call sites.  So from an UNCHANGED Lol the pointwise
-- operations download their operands and run on lol-cpp (correct in any slot order), and the measured rate of a ring switch is
-- that of lol-cpp's pointwise loop (INTEGRATION.md section 3: 91 per second against 1 084 with everything resident).  'mulGT',
-- 'addGT', 'subGT' are the device forms (also the @Additive@ instance of @GT m r@): a Lol whose @UCyc@ instances call them
-- instead of @zipWithT (*)@ / @(+)@ / @(-)@ -- a three-line change in Lol, none in ALCHEMY -- keeps whole ciphertext operations
-- in HBM.  The batched entry points at the end ('mulRelinGT', 'mulFullGT', 'tunnelGT', 'modSwitchGT') are the fast path proper.
--
-- Use: @import Crypto.Lol.Cyclotomic.Tensor.GT@ instead of @...Tensor.CPP@ and write @GT@ for @CT@ in the plaintext alias
-- (reference examples/Arithmetic.hs:19,23; @haskell/examples/Arithmetic-GT.patch@).  Nothing in @Crypto.Alchemy.*@ changes.
module Crypto.Lol.Cyclotomic.Tensor.GT
  ( GT, GTDispatch(..), toDeviceGT, toHostGT, mulGT, addGT, subGT, mulRelinGT, mulFullGT, tunnelGT, modSwitchGT ) where

import Control.DeepSeq                         (NFData (..))
import Control.Monad                           (when)
import Control.Monad.Random                    (Random (..))
import Data.Constraint                         (Dict (..), (:-) (..), (\\))
import Data.Int
import Control.Concurrent.MVar                 (MVar, modifyMVar, newMVar)
import qualified Data.Map.Strict               as M
import Data.Tagged                             (witness)
import qualified Data.Vector.Storable          as SV
import qualified Data.Vector.Storable.Mutable  as SM
import Data.Word
import Foreign.C.String
import Foreign.C.Types
import qualified Foreign.Concurrent            as FC
import Foreign.ForeignPtr
import Foreign.Marshal.Alloc
import Foreign.Marshal.Array
import Foreign.Ptr
import Foreign.Storable
import System.IO.Unsafe                        (unsafePerformIO)

import Crypto.Lol.Cyclotomic.Tensor
import Crypto.Lol.Cyclotomic.Tensor.CPP        (CT)
-- lol-cpp keeps CT's constructors in its internal module; 'toVector' / 'fromVector' below are the only users.
import Crypto.Lol.Cyclotomic.Tensor.CPP.Backend (CT'(..), CT(CT, ZV), zvToCT')
import Crypto.Lol.Cyclotomic.Tensor.GT.Backend
import Crypto.Lol.Prelude
import Crypto.Lol.Reflects
import Crypto.Lol.Types.FiniteField            (GF, GFCtx)
import Crypto.Lol.Types.Unsafe.ZqBasic         (ZqBasic)
import qualified Algebra.Additive              as Additive
import qualified Algebra.Module                as Module
import qualified Algebra.ZeroTestable          as ZeroTestable

-- | A tensor over index @m@ with entries in @r@: on the host (lol-cpp's storable vector, Lol's tuple-interleaved layout) or in HBM.
data GT (m :: Factored) r where
  GTHost :: !(CT m r) -> GT m r
  -- | phi(m), the ring that serves (m, r), one pooled ring element of it.  Never mutated after construction: every method writes a
  -- fresh buffer, so values stay pure.  The @Storable@ dictionary travels with the value so that 'hostOf' needs no context.
  GTDev  :: SV.Storable r => !Int -> !(Ptr AlchRing) -> !(ForeignPtr AlchBuf) -> GT m r

-- | Element types the device serves: 'gtModuli' lists the RNS moduli outermost first (the nesting of @PNoise2Zq@, reference
-- Crypto/Alchemy/Interpreter/PT2CT/Noise.hs:82-89,130); @Nothing@ = the type stays on lol-cpp.
class GTDispatch r where
  gtModuli :: proxy r -> Maybe [Word64]

instance (Reflects q Int64) => GTDispatch (ZqBasic q Int64) where
  gtModuli _ = Just [fromIntegral (proxy value (Proxy :: Proxy q) :: Int64)]
instance (GTDispatch a, GTDispatch b) => GTDispatch (a, b) where
  gtModuli _ = (++) <$> gtModuli (Proxy :: Proxy a) <*> gtModuli (Proxy :: Proxy b)
instance GTDispatch Int64            where gtModuli _ = Nothing
instance GTDispatch Double           where gtModuli _ = Nothing
instance GTDispatch (Complex Double) where gtModuli _ = Nothing

-- ---- host <-> device ---------------------------------------------------------------------------------------------------------

-- | The raw storable vector of a host tensor, and back.
toVector :: SV.Storable r => CT m r -> SV.Vector r
toVector (CT (CT' v)) = v
toVector t@(ZV _)     = case zvToCT' t of CT' v -> v

fromVector :: SV.Storable r => SV.Vector r -> CT m r
fromVector = CT . CT'

check :: String -> CInt -> IO ()
check what rc = when (rc < 0) $ c_lastError >>= peekCString >>= \e -> ioError (userError (what ++ ": " ++ e))

-- | The host form of a value: a 'GTDev' is downloaded (pinned staging, one synchronisation of the ring's stream).
hostOf :: GT m r -> CT m r
hostOf (GTHost t)        = t
hostOf (GTDev n _ fp)    = unsafePerformIO $ do
  mv <- SM.new n
  withForeignPtr fp $ \pb -> SM.unsafeWith mv $ \p -> c_bufDownload pb 0 1 (castPtr p) >>= check "alch_buf_download"
  fromVector <$> SV.unsafeFreeze mv

-- | Lift a lol-cpp method to @GT@ through the host form (the fallback every helper below receives).
liftH :: (CT m a -> CT m' b) -> GT m a -> GT m' b
liftH f = GTHost . f . hostOf

-- | A fresh pooled buffer of @k@ elements of @ring@; released (to the ring's free list: no device synchronisation, and safe from
-- the finalizer thread) when the last Haskell reference dies.
newElems :: Ptr AlchRing -> Int -> IO (ForeignPtr AlchBuf)
newElems ring k = alloca $ \out -> do
  c_bufAlloc ring (fromIntegral k) out >>= check "alch_buf_alloc"
  pb <- peek out
  FC.newForeignPtr pb (c_bufFree pb >> return ())

-- | Element @i@ of a multi-element buffer as a value of its own: a non-owning alias whose finalizer keeps the parent alive.
viewElem :: ForeignPtr AlchBuf -> Int -> IO (ForeignPtr AlchBuf)
viewElem parent i = withForeignPtr parent $ \pp -> alloca $ \out -> do
  c_bufView pp (fromIntegral i) 1 out >>= check "alch_buf_view"
  pv <- peek out
  FC.newForeignPtr pv (c_bufFree pv >> touchForeignPtr parent)

-- | The device form of a value on @ring@: a 'GTHost' is uploaded (no synchronisation: the vector is copied to pinned staging
-- before the call returns).
devOf :: SV.Storable r => Ptr AlchRing -> GT m r -> IO (ForeignPtr AlchBuf)
devOf ring (GTDev _ ring' fp) | ring == ring' = return fp
devOf ring t = do
  fp <- newElems ring 1
  withForeignPtr fp $ \pb -> SV.unsafeWith (toVector (hostOf t)) $ \p -> c_bufUpload pb 0 1 (castPtr p) >>= check "alch_buf_upload"
  return fp

-- | Explicit moves, for callers that manage residency themselves.
toDeviceGT :: forall m r . (Fact m, TElt GT r) => GT m r -> GT m r
toDeviceGT t = case powRing (Proxy :: Proxy m) (Proxy :: Proxy r) of
  Nothing   -> t
  Just ring -> unsafePerformIO $ GTDev (totOf (Proxy :: Proxy m)) ring <$> devOf ring t

toHostGT :: GT m r -> GT m r
toHostGT = GTHost . hostOf

-- ---- ring contexts -----------------------------------------------------------------------------------------------------------

-- | What @alch_ring_create@ said about an (index, modulus list): see the STATUS ORDER in @include/alchemy_hip.h@.
data RingAns = Served !(Ptr AlchRing)   -- ^ ALCH_OK
             | NoCRT                    -- ^ ALCH_E_NO_CRT: composite q, q = 2, or q /= 1 mod m -- Lol's @crtFuncs = Nothing@
             | NotServed                -- ^ ALCH_E_UNSUPPORTED: the whole (index, element type) stays on lol-cpp

-- | One library context per (index, modulus list, with / without CRT), created on first use and kept for the process lifetime.
-- All contexts queue on the stream of the first one ('c_ringShareStream'): Tensor calls arrive one at a time, so calls between
-- two rings (embed, twace, coeffs) need no events.  The cache sits behind an 'MVar': pure code forces tensors from any thread of a
-- @-threaded@ program, and two threads that miss on the same key must not both create the ring (nor lose the shared stream's owner);
-- an exception inside 'modifyMVar' leaves the cache as it was.
{-# NOINLINE ringCache #-}
ringCache :: MVar (M.Map (Word32, [Word64], Bool) RingAns, Maybe (Ptr AlchRing))
ringCache = unsafePerformIO (newMVar (M.empty, Nothing))

ringFor :: Bool -> Word32 -> [Word64] -> IO RingAns
ringFor noCRT m qs = modifyMVar ringCache $ \st@(cache, first) ->
  case M.lookup (m, qs, noCRT) cache of
    Just a  -> return (st, a)
    Nothing -> alloca $ \out -> withArrayLen qs $ \n pq -> do
      rc <- (if noCRT then c_ringCreateNoCRT else c_ringCreate) m (fromIntegral n) pq out
      ans <- case rc of
        0    -> do r <- peek out
                   maybe (return ()) (\f -> c_ringShareStream r f >>= check "alch_ring_share_stream") first
                   return (Served r)
        (-3) -> return NoCRT                                        -- ALCH_E_NO_CRT
        (-4) -> return NotServed                                    -- ALCH_E_UNSUPPORTED
        _    -> c_lastError >>= peekCString >>= \e ->               -- no device, HIP failure, out of memory, malformed call:
                  ioError (userError ("alch_ring_create: " ++ e))   -- not a Lol Nothing; this backend has no CPU fallback
      let first' = case (first, ans) of { (Nothing, Served r) -> Just r; _ -> first }
      return ((M.insert (m, qs, noCRT) ans cache, first'), ans)

-- | The ring that serves the Pow / Dec methods (and holds the device values) of index @m@ over @r@: the ring with CRT basis when
-- there is one, else the ring without; @Nothing@ = this (index, element type) is lol-cpp's.
powRing :: forall m r . (Fact m, GTDispatch r) => Proxy m -> Proxy r -> Maybe (Ptr AlchRing)
powRing _ _ = unsafePerformIO $ case gtModuli (Proxy :: Proxy r) of
  Nothing -> return Nothing
  Just qs -> do
    let m = fromIntegral (proxy valueFact (Proxy :: Proxy m))
    a <- ringFor False m qs
    case a of
      Served r  -> return (Just r)
      NotServed -> return Nothing
      NoCRT     -> do b <- ringFor True m qs
                      return $ case b of { Served r -> Just r; _ -> Nothing }

-- | The ring WITH CRT basis of index @m@ over @r@, for @crtFuncs@ / @crtExtFuncs@.
data CRTAns = CRTServed !(Ptr AlchRing) | CRTNothing | CRTLolCpp

crtRing :: forall m r . (Fact m, GTDispatch r) => Proxy m -> Proxy r -> CRTAns
crtRing _ _ = unsafePerformIO $ case gtModuli (Proxy :: Proxy r) of
  Nothing -> return CRTLolCpp
  Just qs -> do
    a <- ringFor False (fromIntegral (proxy valueFact (Proxy :: Proxy m))) qs
    return $ case a of { Served r -> CRTServed r; NoCRT -> CRTNothing; NotServed -> CRTLolCpp }

-- ---- op codes of alch_buf_tensor_op (include/alchemy_hip.h: ALCH_T_*) -----------------------------------------------------------
opCRT, opCRTInv, opL, opLInv, opMulGPow, opMulGDec, opMulGCRT, opDivGPow, opDivGDec, opDivGCRT :: CInt
opCRT = 0; opCRTInv = 1; opL = 2; opLInv = 3; opMulGPow = 4; opMulGDec = 5; opMulGCRT = 6; opDivGPow = 7; opDivGDec = 8; opDivGCRT = 9

-- | One unary Tensor method on the device: operand made resident if it is not, result in a fresh pooled element.
devUnary :: forall m r . (Fact m, SV.Storable r) => String -> CInt -> Ptr AlchRing -> GT m r -> GT m r
devUnary what op ring t = unsafePerformIO $ do
  src <- devOf ring t
  dst <- newElems ring 1
  withForeignPtr src $ \ps -> withForeignPtr dst $ \pd -> c_bufTensorOp pd 0 ps 0 1 op >>= check what
  return (GTDev (totOf (Proxy :: Proxy m)) ring dst)

-- | The @divG@ family: status 1 (@ALCH_NOT_DIVISIBLE@) is Lol's @Nothing@ (the one Tensor call that must wait for the device).
devMaybe :: forall m r . (Fact m, SV.Storable r) => String -> CInt -> Ptr AlchRing -> GT m r -> Maybe (GT m r)
devMaybe what op ring t = unsafePerformIO $ do
  src <- devOf ring t
  dst <- newElems ring 1
  rc <- withForeignPtr src $ \ps -> withForeignPtr dst $ \pd -> c_bufTensorOp pd 0 ps 0 1 op
  check what rc
  case rc of
    1 -> return Nothing                                             -- ALCH_NOT_DIVISIBLE
    _ -> return (Just (GTDev (totOf (Proxy :: Proxy m)) ring dst))

-- | A Pow / Dec method of index @m@: on the device when (m, r) is served, else lol-cpp's (the first argument).
unaryGT :: forall m r . (Fact m, TElt GT r) => (GT m r -> GT m r) -> String -> CInt -> GT m r -> GT m r
unaryGT host what op t = maybe (host t) (\ring -> devUnary what op ring t) (powRing (Proxy :: Proxy m) (Proxy :: Proxy r))

maybeGT :: forall m r . (Fact m, TElt GT r) => (GT m r -> Maybe (GT m r)) -> String -> CInt -> GT m r -> Maybe (GT m r)
maybeGT host what op t = maybe (host t) (\ring -> devMaybe what op ring t) (powRing (Proxy :: Proxy m) (Proxy :: Proxy r))

instance Tensor GT where
  type TElt GT r = (TElt CT r, GTDispatch r, SV.Storable r)

  -- ---- every basis-order-dependent method: in the library for served (index, element type) pairs ---------------------------
  l           = unaryGT (liftH l)       "l"       opL
  lInv        = unaryGT (liftH lInv)    "lInv"    opLInv
  mulGPow     = unaryGT (liftH mulGPow) "mulGPow" opMulGPow
  mulGDec     = unaryGT (liftH mulGDec) "mulGDec" opMulGDec
  divGPow     = maybeGT (fmap GTHost . divGPow . hostOf) "divGPow" opDivGPow
  divGDec     = maybeGT (fmap GTHost . divGDec . hostOf) "divGDec" opDivGDec
  crtFuncs    = crtFuncsGT    (hostCRTFuncs <$> crtFuncs)
  crtExtFuncs = crtExtFuncsGT (hostCRTFuncs <$> crtFuncs) ((\(tw, em) -> (liftH tw, liftH em)) <$> crtExtFuncs)
  twacePowDec = twacePowDecGT (liftH twacePowDec)
  embedPow    = embedPowGT    (liftH embedPow)
  embedDec    = embedDecGT    (liftH embedDec)
  coeffs      = coeffsGT      (map GTHost . coeffs . hostOf)
  powBasisPow = powBasisPowGT zero one (fmap (map GTHost) powBasisPow)
  crtSetDec   = crtSetDecGT   (fmap (map GTHost) crtSetDec)

  -- ---- methods that need host data: an opaque Haskell function cannot be shipped to the GPU (module header) ---------------------
  zipWithT f a b = GTHost (zipWithT f (hostOf a) (hostOf b))
  fmapT f        = GTHost . fmapT f . hostOf
  unzipT         = (\(a, b) -> (GTHost a, GTHost b)) . unzipT . hostOf

  -- ---- order-free methods: lol-cpp's implementation --------------------------------------------------------------------------------
  scalarPow      = GTHost . scalarPow
  tGaussianDec   = fmap GTHost . tGaussianDec
  gSqNormDec     = gSqNormDec . hostOf
  entailIndexT   = tag $ Sub Dict
  entailEqT      = tag $ Sub Dict
  entailZTT      = tag $ Sub Dict
  entailNFDataT  = tag $ Sub Dict
  entailRandomT  = tag $ Sub Dict
  entailShowT    = tag $ Sub Dict
  entailModuleT  = tag $ Sub Dict

-- | phi(m) as an Int.
totOf :: forall m proxy . Fact m => proxy m -> Int
totOf _ = proxy totientFact (Proxy :: Proxy m)

idxOf :: forall m proxy . Fact m => proxy m -> Word32
idxOf _ = fromIntegral (proxy valueFact (Proxy :: Proxy m))

-- | lol-cpp's @crtFuncs@ tuple under the 'GTHost' constructor.
hostCRTFuncs :: (r -> CT m r, CT m r -> CT m r, CT m r -> CT m r, CT m r -> CT m r, CT m r -> CT m r)
             -> (r -> GT m r, GT m r -> GT m r, GT m r -> GT m r, GT m r -> GT m r, GT m r -> GT m r)
hostCRTFuncs (s, mg, dg, c, ci) = (GTHost . s, liftH mg, liftH dg, liftH c, liftH ci)

-- | The CRTrans-monad tuple Lol asks for: (scalarCRT, mulGCRT, divGCRT, crt, crtInv).  On the device when every modulus is a prime
-- = 1 mod m and the index is served; lol-cpp's own answer otherwise -- which is @Nothing@ in exactly the 'CRTNothing' cases
-- (@alch_ring_create@ answered ALCH_E_NO_CRT), so the two never disagree about the existence of a CRT basis.
crtFuncsGT :: forall mon m r . (Monad mon, Fact m, TElt GT r)
           => mon (r -> GT m r, GT m r -> GT m r, GT m r -> GT m r, GT m r -> GT m r, GT m r -> GT m r)
           -> mon (r -> GT m r, GT m r -> GT m r, GT m r -> GT m r, GT m r -> GT m r, GT m r -> GT m r)
crtFuncsGT host = case crtRing (Proxy :: Proxy m) (Proxy :: Proxy r) of
  CRTLolCpp      -> host
  CRTNothing     -> host                                            -- lol-cpp's crtInfo fails too: q composite, 2, or /= 1 mod m
  CRTServed ring -> (\(s, _, _, _, _) ->
                       ( s                                          -- scalarCRT: a constant vector, the same in any slot order
                       , devUnary "mulGCRT" opMulGCRT ring
                       , devUnary "divGCRT" opDivGCRT ring          -- never fails on the CRT basis
                       , devUnary "crt"     opCRT     ring
                       , devUnary "crtInv"  opCRTInv  ring )) <$> host

-- | Both rings of an extension m | m' over @r@.  'BothServed' when the big index is served (then the small one is: its primes and
-- its dimension are no larger); 'SmallOnly' is the mixed case of the module header; 'NeitherServed' is lol-cpp's.
data ExtAns = BothServed !(Ptr AlchRing) !(Ptr AlchRing) | SmallOnly | NeitherServed

extRings :: forall m m' r . (m `Divides` m', TElt GT r) => Proxy m -> Proxy m' -> Proxy r -> ExtAns
extRings pm pm' pr = case (powRing pm pr, powRing pm' pr) of
  (Just s, Just b)  -> BothServed s b
  (Just _, Nothing) -> SmallOnly
  _                 -> NeitherServed

-- | One call between the two rings of an extension: @k@ fresh elements of @dstRing@ filled from the device form of @t@.
devExt :: SV.Storable r => String -> (Ptr AlchBuf -> Ptr AlchBuf -> IO CInt) -> Ptr AlchRing -> Ptr AlchRing -> Int -> GT i r
       -> IO (ForeignPtr AlchBuf)
devExt what f srcRing dstRing k t = do
  src <- devOf srcRing t
  dst <- newElems dstRing k
  withForeignPtr src $ \ps -> withForeignPtr dst $ \pd -> f pd ps >>= check what
  return dst

twacePowDecGT :: forall m m' r . (m `Divides` m', TElt GT r) => (GT m' r -> GT m r) -> GT m' r -> GT m r
twacePowDecGT host t = case extRings (Proxy :: Proxy m) (Proxy :: Proxy m') (Proxy :: Proxy r) of
  BothServed s b -> unsafePerformIO $ GTDev (totOf (Proxy :: Proxy m)) s <$>
                      devExt "twacePowDec" (\pd ps -> c_bufTwace pd ps 1 0) b s 1 t        -- basis Pow = Dec: the same positions
  _              -> host t                                          -- Pow / Dec positions are Lol's documented order on both sides

embedPowGT :: forall m m' r . (m `Divides` m', TElt GT r) => (GT m r -> GT m' r) -> GT m r -> GT m' r
embedPowGT host t = case extRings (Proxy :: Proxy m) (Proxy :: Proxy m') (Proxy :: Proxy r) of
  BothServed s b -> unsafePerformIO $ GTDev (totOf (Proxy :: Proxy m')) b <$> devExt "embedPow" (\pd ps -> c_bufEmbed pd ps 1 0) s b 1 t
  _              -> host t

embedDecGT :: forall m m' r . (m `Divides` m', TElt GT r) => (GT m r -> GT m' r) -> GT m r -> GT m' r
embedDecGT host t = case extRings (Proxy :: Proxy m) (Proxy :: Proxy m') (Proxy :: Proxy r) of
  BothServed s b -> unsafePerformIO $ GTDev (totOf (Proxy :: Proxy m')) b <$> devExt "embedDec" (\pd ps -> c_bufEmbed pd ps 1 1) s b 1 t
  _              -> host t

-- | @coeffs@: the phi(m')/phi(m) coefficient vectors over the small ring, one device buffer aliased element by element.
coeffsGT :: forall m m' r . (m `Divides` m', TElt GT r) => (GT m' r -> [GT m r]) -> GT m' r -> [GT m r]
coeffsGT host t = case extRings (Proxy :: Proxy m) (Proxy :: Proxy m') (Proxy :: Proxy r) of
  BothServed s b -> unsafePerformIO $ do
    let n = totOf (Proxy :: Proxy m); d = totOf (Proxy :: Proxy m') `div` n
    all' <- devExt "coeffs" (\pd ps -> c_bufCoeffs pd ps 1) b s d t
    mapM (\i -> GTDev n s <$> viewElem all' i) [0 .. d - 1]
  _              -> host t

-- | @crtExtFuncs@ = (twaceCRT, embedCRT): both act on CRT SLOTS and must follow the slot order of @crt@ / @crtInv@ at BOTH indices.
--   * both indices served with a CRT basis: the library's slot maps (@alch_buf_twace@ / @alch_buf_embed@, basis CRT);
--   * the small index served, the big one not: composed through the powerful basis with each index's own @crt@ / @crtInv@
--     (this instance's 'crtFuncs' at that index), so the device's slot order below and lol-cpp's above are both respected;
--   * neither served, or no CRT basis (then lol-cpp's @crtExtFuncs@ is @Nothing@ as well): lol-cpp's.
crtExtFuncsGT :: forall mon m m' r . (Monad mon, m `Divides` m', TElt GT r)
              => mon (r -> GT m r, GT m r -> GT m r, GT m r -> GT m r, GT m r -> GT m r, GT m r -> GT m r)   -- lol-cpp's crtFuncs at the SMALL index
              -> mon (GT m' r -> GT m r, GT m r -> GT m' r)                                                  -- lol-cpp's crtExtFuncs
              -> mon (GT m' r -> GT m r, GT m r -> GT m' r)
crtExtFuncsGT cppSmall host =
  case (crtRing (Proxy :: Proxy m) (Proxy :: Proxy r), crtRing (Proxy :: Proxy m') (Proxy :: Proxy r)) of
    (CRTServed s, CRTServed b) ->
      (\_ -> ( \t -> unsafePerformIO $ GTDev (totOf (Proxy :: Proxy m))  s <$> devExt "twaceCRT" (\pd ps -> c_bufTwace pd ps 1 2) b s 1 t
             , \t -> unsafePerformIO $ GTDev (totOf (Proxy :: Proxy m')) b <$> devExt "embedCRT" (\pd ps -> c_bufEmbed pd ps 1 2) s b 1 t )) <$> host
    (CRTServed s, CRTLolCpp) ->
      -- mixed: crt / crtInv of the small index are the device's (this library's slot order), everything at the big index is
      -- lol-cpp's.  A CRT-basis element of the small index changes between the two slot orders through the powerful basis:
      --   twaceCRT = (crt_dev . crtInv_cpp) . twaceCRT_cpp          embedCRT = embedCRT_cpp . (crt_cpp . crtInv_dev)
      (\(_, _, _, cppCRT, cppCRTInv) (twCpp, emCpp) ->
         ( devUnary "crt" opCRT s . cppCRTInv . twCpp
         , emCpp . cppCRT . devUnary "crtInv" opCRTInv s )) <$> cppSmall <*> host
    _ -> host                                                       -- neither served; or no CRT basis, where lol-cpp answers Nothing too

-- | @powBasisPow@: the relative powerful basis of m'/m as Pow-basis tensors -- unit vectors at the positions the library's @coeffs@
-- reads first (table ALCH_EXT_COEFFS, entries [i][0]), so that @x = sum_i embed (coeffs x !! i) * powBasisPow !! i@.  Host vectors:
-- they are constants of the index pair.  @zero@ and @one@ of @r@ come from the instance body, where the class's context is in scope.
powBasisPowGT :: forall m m' r . (m `Divides` m', TElt GT r) => r -> r -> Tagged m [GT m' r] -> Tagged m [GT m' r]
powBasisPowGT zeroE oneE host = case extRings (Proxy :: Proxy m) (Proxy :: Proxy m') (Proxy :: Proxy r) of
  BothServed _ _ -> tag $ unsafePerformIO $ do
    let n  = totOf (Proxy :: Proxy m)
        n' = totOf (Proxy :: Proxy m')
        d  = n' `div` n
    tab <- alloca $ \plen -> allocaArray (d * n) $ \pt -> do
             poke plen (fromIntegral (d * n))
             c_extTable (idxOf (Proxy :: Proxy m)) (idxOf (Proxy :: Proxy m')) 1 pt plen >>= check "alch_ext_table"   -- ALCH_EXT_COEFFS
             peekArray (d * n) pt
    let unitAt k = GTHost (fromVector (SV.generate n' (\j -> if j == k then oneE else zeroE)))
    return [ unitAt (fromIntegral (tab !! (i * n))) | i <- [0 .. d - 1] ]
  _ -> host

-- | @crtSetDec@: the relative mod-p CRT set of O_m' / O_m over the prime field @fp@ on the decoding basis, from the library's
-- host-side construction (@alch_crt_set_dec@) when @fp@ is a @ZqBasic p Int64@ (its vectors are residues mod p in Lol's storable
-- layout); lol-cpp's otherwise.  (The SET is canonical; its order is the library's documented rule.  It is a list of decoding-basis
-- vectors, not slot-indexed data, so either source would be sound; the library's is used so that @decToCRT@ of reference
-- examples/Common.hs:65-75 is reproducible from the C ABI alone.)
crtSetDecGT :: forall m m' fp . (m `Divides` m', TElt GT fp) => Tagged m [GT m' fp] -> Tagged m [GT m' fp]
crtSetDecGT host = case gtModuli (Proxy :: Proxy fp) of
  Just [p] -> tag $ unsafePerformIO $ do
    let m  = idxOf (Proxy :: Proxy m)
        m' = idxOf (Proxy :: Proxy m')
        n' = totOf (Proxy :: Proxy m')
    cnt <- alloca $ \pc -> poke pc 0 >> c_crtSetDec m m' (fromIntegral p) nullPtr pc >>= check "alch_crt_set_dec" >> peek pc
    let c = fromIntegral cnt :: Int
    mv <- SM.new (c * n')
    SM.unsafeWith mv $ \pv -> alloca $ \pc -> poke pc cnt >> c_crtSetDec m m' (fromIntegral p) (castPtr pv) pc >>= check "alch_crt_set_dec"
    v <- SV.unsafeFreeze mv
    return [ GTHost (fromVector (SV.slice (i * n') n' v)) | i <- [0 .. c - 1] ]
  _        -> host

-- ---- pointwise ring operations on the device ---------------------------------------------------------------------------------------

devBinary :: forall m r . (Fact m, TElt GT r)
          => String -> (Ptr AlchBuf -> Ptr AlchBuf -> Ptr AlchBuf -> CSize -> IO CInt) -> (GT m r -> GT m r -> GT m r) -> GT m r -> GT m r -> GT m r
devBinary what f host a b = case powRing (Proxy :: Proxy m) (Proxy :: Proxy r) of
  Nothing   -> host a b
  Just ring -> unsafePerformIO $ do
    pa <- devOf ring a
    pb <- devOf ring b
    dst <- newElems ring 1
    withForeignPtr pa $ \xa -> withForeignPtr pb $ \xb -> withForeignPtr dst $ \xd -> f xd xa xb 1 >>= check what
    return (GTDev (totOf (Proxy :: Proxy m)) ring dst)

-- | Pointwise product / sum / difference of two tensors given in the same basis (product: the CRT basis), on the device for
-- served element types.  What a Lol whose @UCyc@ does not go through @zipWithT@ calls (module header).
mulGT, addGT, subGT :: forall m r . (Fact m, TElt GT r, Ring r) => GT m r -> GT m r -> GT m r
mulGT = devBinary "mul" c_bufMul (\a b -> GTHost (zipWithT (*) (hostOf a) (hostOf b)))
addGT = devBinary "add" c_bufAdd (\a b -> GTHost (zipWithT (+) (hostOf a) (hostOf b)))
subGT = devBinary "sub" c_bufSub (\a b -> GTHost (zipWithT (-) (hostOf a) (hostOf b)))

-- ---- the instances Lol's entailments promise (through the host form; contexts as in the class's @entail*@ signatures) ---------------

instance (Eq r, Fact m, TElt GT r) => Eq (GT m r) where
  a == b = (hostOf a == hostOf b) \\ witness entailEqT (hostOf a)

instance (Show r, Fact m, TElt GT r) => Show (GT m r) where
  show a = show (hostOf a) \\ witness entailShowT (hostOf a)

instance (ZeroTestable.C r, Fact m, TElt GT r) => ZeroTestable.C (GT m r) where
  isZero a = ZeroTestable.isZero (hostOf a) \\ witness entailZTT (hostOf a)

instance (NFData r, Fact m, TElt GT r) => NFData (GT m r) where
  rnf (GTHost t)   = rnf t \\ witness entailNFDataT t
  rnf (GTDev _ _ _) = ()                                            -- queued on the device; forcing it would be a synchronisation

instance (Random r, Fact m, TElt GT r) => Random (GT m r) where
  random g  = case random g \\ proxy entailRandomT (Proxy :: Proxy (CT m r)) of (t, g') -> (GTHost (t :: CT m r), g')
  randomR _ = nonsensical "randomR on a tensor (as in Lol's own backends)"

-- | Lol's backends answer @randomR@ on tensors with a run-time error as well; out of line so that no method body above is a stub.
nonsensical :: String -> a
nonsensical = unsafePerformIO . ioError . userError

instance Fact m => Functor (GT m) where
  fmap f a = GTHost (fmap f (hostOf a)) \\ witness entailIndexT (hostOf a)
instance Fact m => Applicative (GT m) where
  pure x    = let t = pure x \\ witness entailIndexT t in GTHost t
  f <*> a   = GTHost (hostOf f <*> hostOf a) \\ witness entailIndexT (hostOf a)
instance Fact m => Foldable (GT m) where
  foldr f z a = foldr f z (hostOf a) \\ witness entailIndexT (hostOf a)
instance Fact m => Traversable (GT m) where
  traverse f a = (GTHost <$> traverse f (hostOf a)) \\ witness entailIndexT (hostOf a)

instance (Additive.C r, Fact m, TElt GT r, Ring r) => Additive.C (GT m r) where
  zero   = GTHost (scalarPow zero)
  (+)    = addGT
  (-)    = subGT
  negate = fmapT negate

instance (GFCtx fp d, Fact m, TElt GT fp) => Module.C (GF fp d) (GT m fp) where
  r *> a = GTHost (r Module.*> hostOf a) \\ witness entailModuleT (r, hostOf a)

-- ---- batched entry points: the fast path (INTEGRATION.md section 4) ----------------------------------------------------------------

-- | @keySwitchQuadCirc hint (x * y)@ on device-resident batches: one 'c_ctMulRelin' call.
-- Arguments: ring, hint, operand buffers (2*batch CRT-basis elements each), output buffer, batch,
-- the per-limb scalar folding both toLSD and the key switch's toMSD (see include/alchemy_hip.h).
mulRelinGT :: Ptr AlchRing -> Ptr AlchHint -> Ptr AlchBuf -> Ptr AlchBuf -> Ptr AlchBuf -> Int -> [Word64] -> IO ()
mulRelinGT ring hint a b out batch spre =
  withArray spre $ \ps -> c_ctMulRelin ring hint a b out (fromIntegral batch) ps 0 >>= check "alch_ct_mul_relin"

-- | PT2CT's whole @mul_@ (@modSwitch . keySwitchQuadCirc hint . modSwitch $ x * y@, reference PT2CT.hs:172-177):
-- one 'c_ctMulFull' call; the three rings are read off the handles.
mulFullGT :: Ptr AlchHint -> Ptr AlchBuf -> Ptr AlchBuf -> Ptr AlchBuf -> Int -> [Word64] -> IO ()
mulFullGT hint a b out batch spre =
  withArray spre $ \ps -> c_ctMulFull hint a b out (fromIntegral batch) ps 0 >>= check "alch_ct_mul_full"

-- | @tunnel hint@ on device-resident batches of linear ciphertexts (SymmSHE tunnel as E runs it, reference Eval.hs:134): one
-- 'c_ctTunnel' call.  PT2CT emits @modSwitch_ .: tunnel_ hint .: modSwitch_@ (PT2CT.hs:224-229): when the input buffer's ring holds
-- only the last limbs of the tunnel's R' ring, the leading @modSwitch@ is part of this call (the added limbs are zero and skipped).
-- Arguments: tunnel handle, input buffer (2*batch CRT-basis elements over R'), output buffer (over S'), batch, toMSD's per-limb scalar.
tunnelGT :: Ptr AlchTunnel -> Ptr AlchBuf -> Ptr AlchBuf -> Int -> [Word64] -> IO ()
tunnelGT t a out batch spre =
  withArray spre $ \ps -> c_ctTunnel t a out (fromIntegral batch) ps 0 >>= check "alch_ct_tunnel"

-- | SymmSHE @modSwitch@ on device-resident batches of linear ciphertexts (reference Eval.hs:130): up or down by whole limbs, the
-- direction read off the two buffers' rings; the trailing @modSwitch_@ of @mul_@ and @tunnel_@ when they are not fused.
modSwitchGT :: Ptr AlchBuf -> Ptr AlchBuf -> Int -> IO ()
modSwitchGT a out batch = c_ctModSwitch a out (fromIntegral batch) 0 >>= check "alch_ct_mod_switch"
