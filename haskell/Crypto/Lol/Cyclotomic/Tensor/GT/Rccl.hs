{-# LANGUAGE EmptyDataDecls           #-}
{-# LANGUAGE ForeignFunctionInterface #-}

-- | Raw FFI bindings to @libalchemy_rccl.so@: the optional multi-GPU route of the MI355X backend, one @foreign import@ per entry
-- point of @include/alchemy_rccl.h@.  UNCOMPILED SOURCE, like "Crypto.Lol.Cyclotomic.Tensor.GT.Backend";
-- @tests/test_haskell_shim.py@ checks every import below against the header mechanically (C name, arity, argument and result types).
--
-- Ciphertexts are independent (every op of the evaluator is a pure function of single values,
-- Crypto/Alchemy/Interpreter/Eval.hs:41-53), so the data path has no collective.  Two exchanges exist around it: the key-switch /
-- tunnel hints are generated once per circuit (Crypto/Alchemy/Interpreter/KeysHints.hs:101-129) and every GPU needs them
-- ('c_hintBroadcast', before anything is timed), and result ranges may be collected after a pipeline ('c_bufAllGather').
--
-- Two launch models:
--
--   * one Haskell process drives all GPUs of the node: 'c_commInitAll' n; rank r = device r; the buffer arrays hold n entries.
--     Bound (@forkOS@) threads making @safe@ calls can drive the devices side by side: every entry point of the backend makes its
--     ring's device current for the calling OS thread.
--
--   * one process per GPU (one RTS per device, started by @mpirun@ or a shell loop): rank 0 calls 'c_commUniqueId' and ships the
--     'commIdBytes' bytes to the other processes over any channel the application has; every process calls 'c_commInitRank' with
--     the same id, the number of ranks and its own rank, on the device that is current in it.  The buffer arrays then hold ONE
--     entry, the calling process's buffer ('c_commLocal' tells how many a communicator wants).
--
-- All of them return a status (@0@ = OK, @< 0@ = @ALCH_E_*@, message from 'c_rcclLastError'); the collectives are queued on the
-- stream of each buffer's ring, so @c_sync@ and downloads wait for them.  Every call can block on RCCL: all imports are @safe@.
module Crypto.Lol.Cyclotomic.Tensor.GT.Rccl where

import Data.Word
import Foreign.C.String
import Foreign.C.Types
import Foreign.Ptr

import Crypto.Lol.Cyclotomic.Tensor.GT.Backend (AlchBuf)

data AlchComm

-- | @ALCH_COMM_ID_BYTES@: the size of the id handed from rank 0 to the other processes (RCCL's @ncclUniqueId@).
commIdBytes :: Int
commIdBytes = 128

foreign import ccall safe "alch_rccl_last_error" c_rcclLastError :: IO CString
foreign import ccall safe "alch_comm_init_all"   c_commInitAll   :: CInt -> Ptr (Ptr AlchComm) -> IO CInt
foreign import ccall safe "alch_comm_unique_id"  c_commUniqueId  :: Ptr Word8 -> IO CInt
foreign import ccall safe "alch_comm_init_rank"  c_commInitRank  :: CInt -> CInt -> Ptr Word8 -> Ptr (Ptr AlchComm) -> IO CInt
foreign import ccall safe "alch_comm_destroy"    c_commDestroy   :: Ptr AlchComm -> IO CInt
foreign import ccall safe "alch_comm_size"       c_commSize      :: Ptr AlchComm -> Ptr CInt -> IO CInt
foreign import ccall safe "alch_comm_local"      c_commLocal     :: Ptr AlchComm -> Ptr CInt -> Ptr CInt -> IO CInt
foreign import ccall safe "alch_hint_broadcast"  c_hintBroadcast :: Ptr AlchComm -> CInt -> Ptr (Ptr AlchBuf) -> CSize -> CSize -> IO CInt
foreign import ccall safe "alch_buf_all_gather"  c_bufAllGather  :: Ptr AlchComm -> Ptr (Ptr AlchBuf) -> CSize -> CSize -> Ptr (Ptr AlchBuf) -> IO CInt
