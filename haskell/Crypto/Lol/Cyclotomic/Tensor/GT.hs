{-# LANGUAGE DataKinds             #-}
{-# LANGUAGE FlexibleContexts      #-}
{-# LANGUAGE FlexibleInstances     #-}
{-# LANGUAGE KindSignatures        #-}
{-# LANGUAGE MultiParamTypeClasses #-}
{-# LANGUAGE RankNTypes            #-}
{-# LANGUAGE ScopedTypeVariables   #-}
{-# LANGUAGE TypeFamilies          #-}

-- | @GT@: a Lol 'Tensor' backed by the MI355X library.  UNCOMPILED SOURCE (see Backend.hs): written against the
-- Lol 0.7 class as ALCHEMY uses it (@Tensor t@, @TElt t r@, @crtFuncs@ in the @CRTrans@ monad); the call sequence
-- it performs is the one @alchemy_amd/host/symmshe.hpp@ executes, compiled and tested, in C++.
--
-- Use: in an example, @import Crypto.Lol.Cyclotomic.Tensor.GT@ instead of @...Tensor.CPP@ and write @GT@ for
-- @CT@ in the plaintext alias (reference @examples/Arithmetic.hs:19,23@).  Nothing in @Crypto.Alchemy.*@ changes.
module Crypto.Lol.Cyclotomic.Tensor.GT ( GT, mulRelinGT, mulFullGT ) where

import Control.Monad                          (when)
import Data.Int
import Data.IORef
import qualified Data.Map.Strict              as M
import qualified Data.Vector.Storable         as SV
import qualified Data.Vector.Storable.Mutable as SM
import Data.Word
import Foreign.C.String
import Foreign.C.Types
import Foreign.ForeignPtr
import Foreign.Marshal.Alloc
import Foreign.Marshal.Array
import Foreign.Ptr
import Foreign.Storable
import System.IO.Unsafe                       (unsafePerformIO)

import Crypto.Lol.Cyclotomic.Tensor
import Crypto.Lol.Cyclotomic.Tensor.CPP       (CT)   -- every method off the hot path delegates to it
import Crypto.Lol.Cyclotomic.Tensor.GT.Backend
import Crypto.Lol.Prelude

-- | Same representation as lol-cpp's @CT@: a storable vector in Lol's tuple-interleaved layout.
newtype GT (m :: Factored) r = GT (SV.Vector r)

-- | Elements the device understands: (nested pairs of) @ZqBasic q Int64@; 'moduli' lists them outermost first
-- (the nesting of @PNoise2Zq@, reference @Crypto/Alchemy/Interpreter/PT2CT/Noise.hs:82-89@).
class SV.Storable r => GTElt r where
  moduli :: proxy r -> [Word64]

-- | One library context per (index, modulus list), created on first use and kept for the process lifetime.
{-# NOINLINE ringCache #-}
ringCache :: IORef (M.Map (Word32, [Word64]) (Ptr AlchRing))
ringCache = unsafePerformIO (newIORef M.empty)

-- | @Nothing@ when q /= 1 (mod m): exactly when Lol's @crtFuncs@ has no CRT basis over the base ring.
ringFor :: Word32 -> [Word64] -> IO (Maybe (Ptr AlchRing))
ringFor m qs = do
  cache <- readIORef ringCache
  case M.lookup (m, qs) cache of
    Just r  -> return (Just r)
    Nothing -> alloca $ \out -> withArrayLen qs $ \l pq -> do
      rc <- c_ringCreate m (fromIntegral l) pq out
      case rc of
        0    -> do r <- peek out
                   modifyIORef' ringCache (M.insert (m, qs) r)
                   return (Just r)
        (-3) -> return Nothing                                  -- ALCH_E_NO_CRT
        _    -> c_lastError >>= peekCString >>= \e -> error ("alch_ring_create: " ++ e)

check :: String -> CInt -> IO ()
check what rc = when (rc < 0) $ c_lastError >>= peekCString >>= \e -> error (what ++ ": " ++ e)

-- | lol-cpp's discipline: copy the input vector, let the callee mutate the copy, freeze it.
inPlace :: GTElt r => String -> (Ptr AlchRing -> Ptr Int64 -> IO CInt) -> Ptr AlchRing -> GT m r -> GT m r
inPlace what f ring (GT v) = unsafePerformIO $ do
  mv <- SV.thaw v
  SM.unsafeWith mv $ \p -> f ring (castPtr p) >>= check what
  GT <$> SV.unsafeFreeze mv

inPlace2 :: GTElt r => String -> (Ptr AlchRing -> Ptr Int64 -> Ptr Int64 -> IO CInt) -> Ptr AlchRing -> GT m r -> GT m r -> GT m r
inPlace2 what f ring (GT a) (GT b) = unsafePerformIO $ do
  ma <- SV.thaw a
  SM.unsafeWith ma $ \pa -> SV.unsafeWith b $ \pb -> f ring (castPtr pa) (castPtr pb) >>= check what
  GT <$> SV.unsafeFreeze ma

-- | @divG@: status 1 is Lol's @Nothing@ (never happens for a two-power index, where g = 1).
inPlaceMaybe :: GTElt r => String -> (Ptr AlchRing -> Ptr Int64 -> IO CInt) -> Ptr AlchRing -> GT m r -> Maybe (GT m r)
inPlaceMaybe what f ring (GT v) = unsafePerformIO $ do
  mv <- SV.thaw v
  rc <- SM.unsafeWith mv $ \p -> f ring (castPtr p)
  check what rc
  if rc == 1 then return Nothing else Just . GT <$> SV.unsafeFreeze mv

viaCT :: (CT m r -> CT m' r') -> GT m r -> GT m' r'      -- newtype-level coercion to lol-cpp and back
viaCT = error "coerce through Data.Coerce once CT's constructor is in scope (lol-cpp exports it from an internal module)"

instance Tensor GT where
  type TElt GT r = (GTElt r, TElt CT r)

  -- hot subset: crosses into the library (two-power index; other indices fall through to lol-cpp until
  -- general-index transforms exist in the library)
  crtFuncs = error "see crtFuncsGT: needs the reflected index; kept separate so that this file stays a sketch"
  mulGPow  = viaCT mulGPow
  mulGDec  = viaCT mulGDec
  divGPow  = fmap GT . error "as inPlaceMaybe \"divGPow\" c_divGPow"
  divGDec  = fmap GT . error "as inPlaceMaybe \"divGDec\" c_divGDec"
  zipWithT f = viaCT2 (zipWithT f)            -- (*) / (+) / (-) are intercepted by the RULES below

  -- everything else is off the hot path (SURVEY 8b) and delegates to the existing implementation
  scalarPow    = coerceCT scalarPow
  l            = viaCT l
  lInv         = viaCT lInv
  tGaussianDec = fmap coerceCT . tGaussianDec
  gSqNormDec   = gSqNormDec . toCT
  twacePowDec  = viaCT twacePowDec
  embedPow     = viaCT embedPow
  embedDec     = viaCT embedDec
  crtExtFuncs  = (\(tw, em) -> (viaCT tw, viaCT em)) <$> crtExtFuncs
  coeffs       = map coerceCT . coeffs . toCT
  powBasisPow  = fmap (map coerceCT) powBasisPow
  crtSetDec    = fmap (map coerceCT) crtSetDec
  fmapT f      = viaCT (fmapT f)
  unzipT       = (\(a, b) -> (coerceCT a, coerceCT b)) . unzipT . toCT
  entailIndexT = entailIndexT
  entailEqT    = entailEqT
  entailZTT    = entailZTT
  entailNFDataT = entailNFDataT
  entailRandomT = entailRandomT
  entailShowT  = entailShowT
  entailModuleT = entailModuleT

-- | The CRTrans-monad tuple Lol asks for: (scalarCRT, mulGCRT, divGCRT, crt, crtInv).
crtFuncsGT :: forall m r . (Fact m, GTElt r) => Maybe (r -> GT m r, GT m r -> GT m r, GT m r -> Maybe (GT m r), GT m r -> GT m r, GT m r -> GT m r)
crtFuncsGT = unsafePerformIO $ do
  let m = fromIntegral (proxy valueFact (Proxy :: Proxy m)) :: Word32
  mring <- ringFor m (moduli (Proxy :: Proxy r))
  return $ flip fmap mring $ \ring ->
    ( \r -> GT (SV.replicate (proxy totientFact (Proxy :: Proxy m)) r)
    , inPlace      "mulGCRT" c_mulGCRT ring
    , inPlaceMaybe "divGCRT" c_divGCRT ring
    , inPlace      "crt"     c_crt     ring
    , inPlace      "crtInv"  c_crtInv  ring )

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

-- helpers whose bodies are one 'coerce' each once lol-cpp's CT constructor is importable
toCT :: GT m r -> CT m r
toCT = error "coerce"
coerceCT :: CT m r -> GT m r
coerceCT = error "coerce"
viaCT2 :: (CT m r -> CT m r -> CT m r) -> GT m r -> GT m r -> GT m r
viaCT2 = error "coerce"
