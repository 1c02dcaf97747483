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
{-# LANGUAGE UndecidableInstances  #-}

-- | @GT@: a Lol 'Tensor' whose hot methods run on the MI355X library (@include/alchemy_hip.h@).
--
-- UNCOMPILED SOURCE: no Haskell toolchain (and no Lol) exists in the pipeline that produced this file, so it is
-- written against the Lol 0.7 @Tensor@ class from its published interface and checked mechanically only
-- (@tests/test_haskell_shim.py@: every method of the class is defined here, none is an @error@ stub, and every
-- foreign symbol used exists in @GT/Backend.hs@ with the header's signature).  The same call sequence, compiled
-- and tested, is @alchemy_amd/host/symmshe.hpp@ (C++).
--
-- Design.  @GT m r@ is a newtype over lol-cpp's @CT m r@: every method that is not on ALCHEMY's hot path is
-- lol-cpp's, reached by 'coerce'.  The hot methods -- @crt@, @crtInv@, @mulG*@, @divG*@, @l@, @lInv@ and
-- @zipWithT@ for @(*)@ / @(+)@ (reference call sites: @(*)@ on @CT@, @modSwitch@, @keySwitchQuadCirc@,
-- Crypto/Alchemy/Interpreter/Eval.hs:65-67,130,133) -- cross into the library whenever the element type is a
-- (nested pair of) @ZqBasic q Int64@, any cyclotomic index; other element types (@Double@, @Complex Double@,
-- @Int64@, @RRq@) stay on lol-cpp.
--
-- Use: @import Crypto.Lol.Cyclotomic.Tensor.GT@ instead of @...Tensor.CPP@ and write @GT@ for @CT@ in the
-- plaintext alias (reference examples/Arithmetic.hs:19,23; @haskell/examples/Arithmetic-GT.patch@).  Nothing in
-- @Crypto.Alchemy.*@ changes.
module Crypto.Lol.Cyclotomic.Tensor.GT ( GT, GTDispatch(..), mulRelinGT, mulFullGT, tunnelGT, modSwitchGT ) where

import Control.Monad                          (when)
import Data.Coerce                            (coerce)
import Data.Int
import Data.IORef
import qualified Data.Map.Strict              as M
import qualified Data.Vector.Storable         as SV
import qualified Data.Vector.Storable.Mutable as SM
import Data.Word
import Foreign.C.String
import Foreign.C.Types
import Foreign.Marshal.Alloc
import Foreign.Marshal.Array
import Foreign.Ptr
import Foreign.Storable
import System.IO.Unsafe                       (unsafePerformIO)

import Crypto.Lol.Cyclotomic.Tensor
import Crypto.Lol.Cyclotomic.Tensor.CPP       (CT)
-- lol-cpp keeps CT's constructors in its internal module; the two marshalling functions below are the only users.
import Crypto.Lol.Cyclotomic.Tensor.CPP.Backend (CT'(..), CT(CT, ZV), zvToCT')
import Crypto.Lol.Cyclotomic.Tensor.GT.Backend
import Crypto.Lol.Prelude
import Crypto.Lol.Reflects
import Crypto.Lol.Types.Unsafe.ZqBasic        (ZqBasic)

-- | Same representation as lol-cpp's tensor.
newtype GT (m :: Factored) r = GT (CT m r)

-- | Element types the device serves: 'gtModuli' lists the RNS moduli outermost first (the nesting of
-- @PNoise2Zq@, reference Crypto/Alchemy/Interpreter/PT2CT/Noise.hs:82-89,130); @Nothing@ = stay on lol-cpp.
class GTDispatch r where
  gtModuli :: proxy r -> Maybe [Word64]

instance (Reflects q Int64) => GTDispatch (ZqBasic q Int64) where
  gtModuli _ = Just [fromIntegral (proxy value (Proxy :: Proxy q) :: Int64)]
instance (GTDispatch a, GTDispatch b) => GTDispatch (a, b) where
  gtModuli _ = (++) <$> gtModuli (Proxy :: Proxy a) <*> gtModuli (Proxy :: Proxy b)
instance GTDispatch Int64            where gtModuli _ = Nothing
instance GTDispatch Double           where gtModuli _ = Nothing
instance GTDispatch (Complex Double) where gtModuli _ = Nothing

-- | The raw storable vector (Lol's tuple-interleaved layout) of a tensor, and back.
toVector :: SV.Storable r => GT m r -> SV.Vector r
toVector (GT (CT (CT' v))) = v
toVector (GT t@(ZV _))     = case zvToCT' t of CT' v -> v

fromVector :: SV.Storable r => SV.Vector r -> GT m r
fromVector = GT . CT . CT'

-- | One library context per (index, modulus list), created on first use and kept for the process lifetime.
{-# NOINLINE ringCache #-}
ringCache :: IORef (M.Map (Word32, [Word64], Bool) (Ptr AlchRing))
ringCache = unsafePerformIO (newIORef M.empty)

-- | @Nothing@ when q /= 1 (mod m): exactly when Lol's @crtFuncs@ has no CRT basis over the base ring.
-- With @noCRT@ the ring serves the Pow / Dec methods only (@alch_ring_create_nocrt@) and always exists.
ringFor :: Bool -> Word32 -> [Word64] -> IO (Maybe (Ptr AlchRing))
ringFor noCRT m qs = do
  cache <- readIORef ringCache
  case M.lookup (m, qs, noCRT) cache of
    Just r  -> return (Just r)
    Nothing -> alloca $ \out -> withArrayLen qs $ \n pq -> do
      rc <- (if noCRT then c_ringCreateNoCRT else c_ringCreate) m (fromIntegral n) pq out
      case rc of
        0    -> do r <- peek out
                   modifyIORef' ringCache (M.insert (m, qs, noCRT) r)
                   return (Just r)
        (-3) -> return Nothing                                  -- ALCH_E_NO_CRT
        _    -> c_lastError >>= peekCString >>= \e -> error ("alch_ring_create: " ++ e)

-- | The ring for the Pow / Dec methods of index @m@ over @r@: the CRT ring when there is one, else a no-CRT ring.
powRing :: forall m r proxy . (Fact m, GTDispatch r) => proxy (GT m r) -> Maybe (Ptr AlchRing)
powRing _ = unsafePerformIO $ case gtModuli (Proxy :: Proxy r) of
  Nothing -> return Nothing
  Just qs -> do let m = fromIntegral (proxy valueFact (Proxy :: Proxy m))
                mr <- ringFor False m qs
                maybe (ringFor True m qs) (return . Just) mr

check :: String -> CInt -> IO ()
check what rc = when (rc < 0) $ c_lastError >>= peekCString >>= \e -> error (what ++ ": " ++ e)

-- | lol-cpp's discipline: copy the input vector, let the callee mutate the copy, freeze it.
inPlace :: SV.Storable r => String -> (Ptr AlchRing -> Ptr Int64 -> IO CInt) -> Ptr AlchRing -> GT m r -> GT m r
inPlace what f ring t = unsafePerformIO $ do
  mv <- SV.thaw (toVector t)
  SM.unsafeWith mv $ \p -> f ring (castPtr p) >>= check what
  fromVector <$> SV.unsafeFreeze mv

inPlace2 :: SV.Storable r => String -> (Ptr AlchRing -> Ptr Int64 -> Ptr Int64 -> IO CInt) -> Ptr AlchRing -> GT m r -> GT m r -> GT m r
inPlace2 what f ring a b = unsafePerformIO $ do
  ma <- SV.thaw (toVector a)
  SM.unsafeWith ma $ \pa -> SV.unsafeWith (toVector b) $ \pb -> f ring (castPtr pa) (castPtr pb) >>= check what
  fromVector <$> SV.unsafeFreeze ma

-- | The @divG@ family: status 1 (@ALCH_NOT_DIVISIBLE@) is Lol's @Nothing@.
inPlaceMaybe :: SV.Storable r => String -> (Ptr AlchRing -> Ptr Int64 -> IO CInt) -> Ptr AlchRing -> GT m r -> Maybe (GT m r)
inPlaceMaybe what f ring t = unsafePerformIO $ do
  mv <- SV.thaw (toVector t)
  rc <- SM.unsafeWith mv $ \p -> f ring (castPtr p)
  check what rc
  if rc == 1 then return Nothing else Just . fromVector <$> SV.unsafeFreeze mv

-- | Run @dev@ on the device when the element type has moduli, else @host@ (lol-cpp's method under the newtype).
onDevice :: forall m r a . (Fact m, GTDispatch r) => GT m r -> (Ptr AlchRing -> a) -> a -> a
onDevice t dev host = maybe host dev (powRing (Just t))

instance Tensor GT where
  type TElt GT r = (TElt CT r, GTDispatch r, SV.Storable r)

  -- ---- hot subset: crosses into the library ------------------------------------------------------------
  l       t = onDevice t (\ring -> inPlace "l"       c_l       ring t) (coerce (l       :: CT m r -> CT m r) t)
  lInv    t = onDevice t (\ring -> inPlace "lInv"    c_lInv    ring t) (coerce (lInv    :: CT m r -> CT m r) t)
  mulGPow t = onDevice t (\ring -> inPlace "mulGPow" c_mulGPow ring t) (coerce (mulGPow :: CT m r -> CT m r) t)
  mulGDec t = onDevice t (\ring -> inPlace "mulGDec" c_mulGDec ring t) (coerce (mulGDec :: CT m r -> CT m r) t)
  divGPow t = onDevice t (\ring -> inPlaceMaybe "divGPow" c_divGPow ring t) (coerce (divGPow :: CT m r -> Maybe (CT m r)) t)
  divGDec t = onDevice t (\ring -> inPlaceMaybe "divGDec" c_divGDec ring t) (coerce (divGDec :: CT m r -> Maybe (CT m r)) t)
  crtFuncs = crtFuncsGT
  -- (*) / (+) / (-) on ring elements reach the device through the rewrite rules at the end of this file;
  -- an arbitrary function cannot be shipped to the GPU
  zipWithT f a b = coerce (zipWithT f (coerce a :: CT m a') (coerce b :: CT m b'))

  -- ---- off the hot path (SURVEY 8b): lol-cpp's implementation under the newtype ---------------------------
  scalarPow     = coerce (scalarPow   :: r -> CT m r)
  tGaussianDec  = fmap GT . tGaussianDec
  gSqNormDec    = gSqNormDec . (coerce :: GT m r -> CT m r)
  twacePowDec   = coerce (twacePowDec :: CT m' r -> CT m r)
  embedPow      = coerce (embedPow    :: CT m r -> CT m' r)
  embedDec      = coerce (embedDec    :: CT m r -> CT m' r)
  crtExtFuncs   = (\(tw, em) -> (coerce tw, coerce em)) <$> (crtExtFuncs :: mon (CT m' r -> CT m r, CT m r -> CT m' r))
  coeffs        = map GT . coeffs . (coerce :: GT m' r -> CT m' r)
  powBasisPow   = fmap (map GT) powBasisPow
  crtSetDec     = fmap (map GT) crtSetDec
  fmapT f       = GT . fmapT f . (coerce :: GT m a -> CT m a)
  unzipT        = (\(a, b) -> (GT a, GT b)) . unzipT . (coerce :: GT m (a, b) -> CT m (a, b))
  entailIndexT  = tag $ Sub Dict
  entailEqT     = tag $ Sub Dict
  entailZTT     = tag $ Sub Dict
  entailNFDataT = tag $ Sub Dict
  entailRandomT = tag $ Sub Dict
  entailShowT   = tag $ Sub Dict
  entailModuleT = tag $ Sub Dict

-- | The CRTrans-monad tuple Lol asks for: (scalarCRT, mulGCRT, divGCRT, crt, crtInv).  On the device when the
-- element type has moduli AND every modulus is 1 mod m (else @alch_ring_create@ answers ALCH_E_NO_CRT, Lol's
-- @Nothing@, and lol-cpp's own 'crtFuncs' decides: it fails in the same cases).
crtFuncsGT :: forall mon m r . (CRTrans mon r, Fact m, TElt GT r)
           => mon (r -> GT m r, GT m r -> GT m r, GT m r -> GT m r, GT m r -> GT m r, GT m r -> GT m r)
crtFuncsGT =
  let m     = fromIntegral (proxy valueFact (Proxy :: Proxy m)) :: Word32
      mring = unsafePerformIO $ maybe (return Nothing) (ringFor False m) (gtModuli (Proxy :: Proxy r))
      host  = (\(s, mg, dg, c, ci) -> (coerce s, coerce mg, coerce dg, coerce c, coerce ci))
                <$> (crtFuncs :: mon (r -> CT m r, CT m r -> CT m r, CT m r -> CT m r, CT m r -> CT m r, CT m r -> CT m r))
  in case mring of
       Nothing   -> host
       Just ring -> (\(s, _, _, _, _) ->
                       ( s
                       , inPlace "mulGCRT" c_mulGCRT ring
                       , inPlace "divGCRT" c_divGCRT ring         -- never fails on the CRT basis
                       , inPlace "crt"     c_crt     ring
                       , inPlace "crtInv"  c_crtInv  ring )) <$> host

-- | Pointwise product / sum / difference of two tensors of the same basis on the device (Cyc's ring operations
-- on the CRT basis arrive here through the rules below).
mulGT, addGT, subGT :: forall m r . (Fact m, TElt GT r, Ring r) => GT m r -> GT m r -> GT m r
mulGT a b = onDevice a (\ring -> inPlace2 "mul" c_mul ring a b) (coerce (zipWithT (*) (coerce a :: CT m r) (coerce b :: CT m r)))
addGT a b = onDevice a (\ring -> inPlace2 "add" c_add ring a b) (coerce (zipWithT (+) (coerce a :: CT m r) (coerce b :: CT m r)))
subGT a b = onDevice a (\ring -> inPlace2 "sub" c_sub ring a b) (coerce (zipWithT (-) (coerce a :: CT m r) (coerce b :: CT m r)))

{-# RULES
"zipWithT/GT/mul" forall (a :: GT m r) b . zipWithT (*) a b = mulGT a b
"zipWithT/GT/add" forall (a :: GT m r) b . zipWithT (+) a b = addGT a b
"zipWithT/GT/sub" forall (a :: GT m r) b . zipWithT (-) a b = subGT a b
  #-}

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
