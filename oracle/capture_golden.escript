#!/usr/bin/env escript
%% -*- erlang -*-
%% TEST INFRASTRUCTURE, to be run by anyone who HAS Erlang/OTP and a checkout of cloudozer/BWT (it cannot run
%% in the authoring container: no erl there).  It drives the REAL reference modules over the committed inputs and
%% writes the golden vectors that tests/test_erlang_golden.py checks the oracle (and through it the GPU path) against.
%%
%%   cd <cloudozer/BWT checkout> && make            (or: erlc -I include -o ebin src/*.erl)
%%   escript <this repo>/oracle/capture_golden.escript <this repo>/tests/golden/erlang
%%
%% Output: erlang_vectors.txt, one line per input line of capture_inputs.txt (strings print as lists of bytes: ~w):
%%   sga Q            -> "sga [{S,D},...]" | "sga crash"                                     sga:sga/6            src/sga.erl:17
%%   seeds Q          -> "seeds no_seeds|too_many_seeds|[..]|crash"                          bwa:find_seeds/6     src/bwa.erl:12
%%   sw Q R           -> "sw no_match|{Score,\"Cigar\"}"                                     sw:sw/2              src/sw.erl:47
%%   align Name Q     -> "align Name nothing|crash|[{{Score,\"Cigar\"},Pos}]"                the Cigars fold      src/worker_bwt.erl:84-97
%%   pick S1,S2,..    -> "pick I"  0-based index (in list order) of the head of lists:sort(fun R1 > R2, Cigars)   src/worker_bwt.erl:99-106
%%   fastq File N1,N2 -> "fastq File T1|T2|.."  the results of successive fastq:read_seqs(Dev, Ni) on fastq/File       src/fastq.erl:20-40
%%   genfs Q Len D R  -> "genfs Md5 [..]|uncompilable"  md5 of the fs.erl seeds:generate_fs/3 writes + fs:find_seeds(R) src/seeds.erl:11-85
%%   line N C P Cg S R -> "line Hex"  the bytes master's io:format prints for a cast          src/master.erl:151-153
main([Dir]) ->
    code:add_patha("ebin"),
    {ok, Bin} = file:read_file(filename:join(Dir, "capture_inputs.txt")),
    {Meta, FM} = fm_index:get_index("GL000193.1", 1),
    {Pc,Pg,Pt,Last} = proplists:get_value(pointers, Meta),
    Shift = proplists:get_value(shift, Meta),
    {ok, Ref} = file:read_file("fm_indices/GL000193.1_p1.ref"),
    Ext = list_to_binary(lists:duplicate(200, $N)),
    Ref1 = <<Ext/binary, Ref/binary, Ext/binary>>,
    put(dir, Dir),
    Lines = string:tokens(binary_to_list(Bin), "\n"),
    Out = [one(string:tokens(L, " "), FM, Pc, Pg, Pt, Last, Shift, Ref1) || L <- Lines],
    ok = file:write_file(filename:join(Dir, "erlang_vectors.txt"), [[O, $\n] || O <- Out]),
    io:format("~b vectors written~n", [length(Out)]).

one(["sga", Q], FM, Pc, Pg, Pt, Last, _, _) ->
    io_lib:format("sga ~w", [catch_crash(fun() -> sga:sga(FM, Pc, Pg, Pt, Last, Q) end)]);
one(["seeds", Q], FM, Pc, Pg, Pt, Last, _, _) ->
    io_lib:format("seeds ~w", [catch_crash(fun() -> bwa:find_seeds(FM, Pc, Pg, Pt, Last, Q) end)]);
one(["sw", Q, R], _, _, _, _, _, _, _) ->
    io_lib:format("sw ~p", [sw:sw(Q, R)]);
one(["align", Name, Q], FM, Pc, Pg, Pt, Last, Shift, Ref1) ->
    R = catch_crash(fun() ->
        case sga:sga(FM, Pc, Pg, Pt, Last, Q) of
            [] -> nothing;
            Seeds ->
                lists:foldl(fun({S, D}, Acc) ->                      %% worker_bwt.erl:84-97
                    RefLen = length(Q) + D,
                    Start = S - RefLen + 200,
                    <<_:Start/bytes, W:RefLen/bytes, _/binary>> = Ref1,
                    case sw:sw(Q, binary_to_list(W)) of
                        no_match -> Acc;
                        Cigar -> [{Cigar, S - RefLen + Shift} | Acc]
                    end
                end, [], Seeds)
        end end),
    io_lib:format("align ~s ~p", [Name, R]);
one(["pick", Scores], _, _, _, _, _, _, _) ->
    %% the candidates carry their list index as Pos; the comparator is the worker's own (worker_bwt.erl:104)
    Ss = [list_to_integer(S) || S <- string:tokens(Scores, ",")],
    Cigars = [{{S, "c"}, I, "r"} || {S, I} <- lists:zip(Ss, lists:seq(0, length(Ss) - 1))],
    [{_, Top, _} | _] = lists:sort(fun({{R1,_},_,_}, {{R2,_},_,_}) -> R1 > R2 end, Cigars),
    io_lib:format("pick ~b", [Top]);
one(["fastq", File, Ns], _, _, _, _, _, _, _) ->
    {ok, Dev} = file:open(filename:join([get(dir), "fastq", File]), []),      %% list mode, as master.erl:119 opens it
    Rs = [catch_crash(fun() -> fastq:read_seqs(Dev, list_to_integer(N)) end) || N <- string:tokens(Ns, ",")],
    file:close(Dev),
    io_lib:format("fastq ~s ~s", [File, string:join([lists:flatten(io_lib:format("~w", [R])) || R <- Rs], "|")]);
one(["genfs", Q, Len, Defects, R], _, _, _, _, _, _, _) ->
    ok = seeds:generate_fs(Q, list_to_integer(Len), list_to_integer(Defects)),          %% writes ./fs.erl
    {ok, Text} = file:read_file("fs.erl"),
    Md5 = lists:flatten([io_lib:format("~2.16.0b", [B]) || <<B>> <= erlang:md5(Text)]),
    Res = case compile:file("fs", [binary, return_errors]) of
              {ok, fs, Code} ->
                  code:purge(fs), {module, fs} = code:load_binary(fs, "fs.beam", Code),
                  io_lib:format("~w", [catch_crash(fun() -> fs:find_seeds(R) end)]);
              _ -> "uncompilable"
          end,
    io_lib:format("genfs ~s ~s", [Md5, Res]);
one(["line", Name, Chrom, Pos, Cigar, Score, RefSeq], _, _, _, _, _, _, _) ->
    Txt = io_lib:format("~n~s      ~s      ~b      ~s      ~b      ~s~n",
                        [Name, Chrom, list_to_integer(Pos), Cigar, list_to_integer(Score), RefSeq]),   %% master.erl:153
    io_lib:format("line ~s", [[io_lib:format("~2.16.0b", [B]) || <<B>> <= iolist_to_binary(Txt)]]).

catch_crash(F) -> try F() catch _:_ -> crash end.
